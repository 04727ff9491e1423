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
