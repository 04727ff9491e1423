"""The split-precision planner's coverage of the hyper-parameter space, on the host (umx_plan_check: no device needed): which models
umx_create(UMX_PREC_DEFAULT) runs on the fast kernels and which fall back -- with a warning -- to the exact-fp32 engine."""
import numpy as np

from unmicst_amd import model, umx


def _random_hps(n, seed):
    """The fuzz set of tests/test_gpu_parity.py::test_forward_tiles_random_hyper_parameters (same generator, same seed there)."""
    rng = np.random.default_rng(seed)
    out = []
    while len(out) < n:
        graph = int(rng.integers(0, 2))
        L = int(rng.integers(1, 6))
        ks = int(rng.choice([3, 3, 5, 7]))
        P = int(rng.choice([16, 32, 64, 128]))
        if P >> L < 2 or (ks == 7 and P >> L < 4):
            continue
        hp = model.HParams(graph, P, int(rng.integers(1, 4)), int(rng.integers(2, 5)), int(rng.integers(3, 25)), L, ks,
                           int(rng.integers(0, 3)) if graph == model.GRAPH_LEGACY else int(rng.integers(0, 2)))
        if hp.flops_per_tile() > 3e9:
            continue
        out.append(hp)
    return out


def test_every_shipped_model_runs_on_the_split_precision_kernels():
    for key, hp in model.KNOWN_HP.items():
        assert umx.plan_check(hp) == "", key


def test_at_least_eight_of_the_fourteen_fuzz_sets_run_on_the_split_precision_kernels():
    """VERDICT r5 item 8.  The three that do not are 7 x 7 filters over 4 x 4-pixel layers: 16 images x (4 + 6)^2 halo pixels exceed the
    1024-pixel halo image of both kernel families' tile geometry."""
    reasons = [umx.plan_check(hp) for hp in _random_hps(14, 2026)]
    assert sum(r == "" for r in reasons) >= 11, reasons
    assert all("halo too large" in r for r in reasons if r), reasons


def test_a_refusal_names_the_layer_and_the_reason():
    hp = model.HParams(model.GRAPH_LEGACY, 32, 1, 2, 13, 3, 7, 1)
    r = umx.plan_check(hp)
    assert r.startswith("lb.conv:") and "halo" in r


def test_the_command_line_tool_reports_both_outcomes(tmp_path):
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "plan_check.py"), "nucleiDAPI1-5", os.path.join(root, "models", "nucleiDAPI")],
                       capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.count("split precision") == 2, r.stdout + r.stderr


def test_the_planners_mx_fp6_packing_is_the_ocp_rule():
    """The F6 form's weight images: blocks of 32 with one e8m0 scale 2^(floor(log2 max) - 2) and e2m3 elements (bias 1, subnormals in
    steps of 1/8, largest 7.5), round-to-nearest-even, saturating, element i in bits [6 i, 6 i + 6) -- against a numpy statement of the
    rule (the device side of the same format is probed on the GPU: tools/probes/mx_fp6_semantics.hip)."""
    import ctypes
    L = umx.load()
    L.umx_test_mx_pack_e2m3.restype = ctypes.c_int
    L.umx_test_mx_pack_e2m3.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    grid = np.array([m / 8 for m in range(8)] + [(1 + m / 8) * 2.0 ** (e - 1) for e in (1, 2, 3) for m in range(8)])   # the 32 magnitudes
    rng = np.random.default_rng(0)
    for trial in range(300):
        scale = 2.0 ** rng.integers(-20, 12)
        v = rng.normal(size=32) * scale * (rng.random(32) < 0.9) * 2.0 ** -rng.integers(0, 6, 32)
        if trial == 0:
            v[:] = 0.0
        if trial == 1:
            v = np.array([7.75, -7.75, 7.5, 7.25, 0.0625, 0.1875, 0.5625, 1.0625] + [0.0] * 24)     # saturation and exact ties
        out = np.zeros(24, np.uint8)
        vv = np.ascontiguousarray(v, np.float64)
        e8 = L.umx_test_mx_pack_e2m3(vv.ctypes.data, out.ctypes.data)
        amax = np.abs(v).max()
        if amax == 0:
            assert not out.any()
            continue
        want_e = int(np.floor(np.log2(amax))) - 2
        assert e8 == want_e + 127
        bits = np.unpackbits(out, bitorder="little")
        codes = np.array([int(sum(int(bits[6 * i + b]) << b for b in range(6))) for i in range(32)])
        a = np.minimum(np.abs(v) / 2.0 ** want_e, 7.5)
        idx = np.array([min(range(32), key=lambda c: (abs(grid[c] - x), c & 1)) for x in a])   # nearest, ties to the even code
        assert np.array_equal(codes & 31, idx), (trial, v, codes)
        assert np.array_equal((codes >> 5)[idx > 0], (v < 0).astype(int)[idx > 0])
