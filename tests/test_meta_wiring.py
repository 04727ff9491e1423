"""The engine's wiring against the op graph the reference SAVED with its models.

tests/golden/meta_graph_<model>.json is the inference path of models/<model>/model.ckpt.meta (tools/make_meta_fixture.py: a
raw protobuf walk, data only).  The checker below replays the engine's launch list (umx_describe_graph: buffers, operand
groups in concat order, fused pool / activation / BatchNorm placement, constants) against that graph: every launch must find
the TensorFlow convolution(s) it stands for -- same source tensors in the same concat order, same filter shape, stride and
padding -- followed by the same BatchNorm / activation / pool chain with the same epsilon and alpha, and every op of the
saved inference graph must be claimed by exactly one launch.  v2 NUMERICS cannot be pinned (no weights are shipped for the
solo / duo models); the wiring can, and this is that pin.  Host only: no GPU, no reference tree at test time."""
import copy
import json
import os

import pytest

from unmicst_amd import model, umx

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = {"nucleiDAPI1-5": "nucleiDAPI1-5", "nucleiDAPILAMIN": "nucleiDAPILAMIN", "nucleiDAPI": "nucleiDAPI"}


def load(name):
    return json.load(open(os.path.join(GOLD, "meta_graph_%s.json" % name)))


def check_wiring(desc, fixture):
    """Raises AssertionError where the engine's launch list and the saved graph disagree."""
    ops = fixture["ops"]
    by = {o["name"]: o for o in ops}
    users = {}
    for o in ops:
        for i in o["inputs"]:
            if isinstance(i, str):
                users.setdefault(i, []).append(o)
    claimed = set()
    tensor = {0: "placeholders/data"}        # engine buffer id -> the TensorFlow op whose output it holds
    claimed.add("placeholders/data")

    def filt(conv, pos=1):
        v = conv["inputs"][pos]
        assert isinstance(v, dict), (conv["name"], "filter is not a variable")
        return v["shape"]

    def only_user(name, what):
        us = [u for u in users.get(name, []) if u["name"] not in claimed]
        assert len(us) >= 1, "%s: nothing consumes %s" % (what, name)
        return us

    def follow_chain(start, L):
        """BatchNorm / activation / pool ops behind `start`, checked against the launch's fused epilogue; returns the last op."""
        cur, seen = start, []
        while True:
            nxt = [u for u in users.get(cur, []) if u["op"] in ("FusedBatchNorm", "FusedBatchNormV3", "LeakyRelu", "Relu", "MaxPool", "Softmax")
                   and u["name"] not in claimed]
            if not nxt:
                break
            assert len(nxt) == 1, (L["name"], "ambiguous epilogue", [n["name"] for n in nxt])
            o = nxt[0]
            seen.append(o)
            claimed.add(o["name"])
            cur = o["name"]
            if o["op"] in ("MaxPool", "Softmax"):
                break
        kinds = [o["op"].replace("V3", "") for o in seen]
        act = {"leaky_relu": "LeakyRelu", "relu": "Relu", "none": None, "softmax": "Softmax"}[L["activation"]]
        want = []
        if L["batch_norm"] == "before_activation":
            want.append("FusedBatchNorm")
        if act:
            want.append(act)
        if L["batch_norm"] == "after_activation":
            want.append("FusedBatchNorm")
        if L["max_pool"]:
            want.append("MaxPool")
        assert kinds == want, (L["name"], "epilogue chain", kinds, "engine fuses", want)
        for o in seen:
            if o["op"].startswith("FusedBatchNorm"):
                assert abs(o["attrs"]["epsilon"] - desc["bn_epsilon"]) < 1e-9 and o["attrs"]["data_format"] == "NHWC", o
            if o["op"] == "LeakyRelu":
                assert abs(o["attrs"]["alpha"] - desc["leaky_slope"]) < 1e-7, o
            if o["op"] == "MaxPool":
                k = L["max_pool"]
                assert o["attrs"]["ksize"] == [1, k, k, 1] and o["attrs"]["strides"] == [1, k, k, 1], o
        return cur

    def conv_of(src_name, ks, cin, cout, what):
        cands = [u for u in users.get(src_name, []) if u["op"] == "Conv2D" and u["name"] not in claimed and filt(u) == [ks, ks, cin, cout]]
        assert cands, "%s: no Conv2D %dx%d %d->%d reads %s" % (what, ks, ks, cin, cout, src_name)
        c = cands[0]
        assert c["attrs"]["strides"] == [1, 1, 1, 1] and c["attrs"]["padding"] == "SAME" and c["attrs"]["data_format"] == "NHWC", c
        claimed.add(c["name"])
        return c

    for L in desc["launches"]:
        g = L["groups"]
        for gi in g:
            assert gi["src"] in tensor, (L["name"], "reads buffer %d before anything wrote it" % gi["src"])
        cout = L["out_channels"]
        if L["kind"] == "conv_transpose":
            src = tensor[g[0]["src"]]
            cands = [u for u in users.get(src, []) if u["op"] == "Conv2DBackpropInput" and u["name"] not in claimed]
            assert cands, (L["name"], "no conv2d_transpose reads", src)
            t = cands[0]
            ks = g[0]["ks"]
            assert filt(t, 0) == [ks, ks, cout, g[0]["channels"]], (L["name"], filt(t, 0))
            s = L["stride"]
            assert t["attrs"]["strides"] == [1, s, s, 1] and t["attrs"]["padding"] == "SAME", t
            claimed.add(t["name"])
            tensor[L["dst"]] = follow_chain(t["name"], L)
            continue
        if len(g) == 2:
            want = [tensor[g[0]["src"]], tensor[g[1]["src"]]]
            cat = [u for u in ops if u["op"] == "ConcatV2" and u["name"] not in claimed and u["inputs"] == want]
            if cat:
                # concat3([skip, us]) -> conv: the groups ARE the concat inputs, in order
                claimed.add(cat[0]["name"])
                c = conv_of(cat[0]["name"], g[0]["ks"], g[0]["channels"] + g[1]["channels"], cout, L["name"])
                tensor[L["dst"]] = follow_chain(c["name"], L)
                continue
            # two convolutions summed (legacy block: last extra conv + 1x1 shortcut of the block input); a launch whose groups
            # are a concat in another order, or from other tensors, finds no such pair and fails here
            a = conv_of(tensor[g[0]["src"]], g[0]["ks"], g[0]["channels"], cout, L["name"])
            b = conv_of(tensor[g[1]["src"]], g[1]["ks"], g[1]["channels"], cout, L["name"])
            adds = [u for u in users.get(a["name"], []) if u["op"] in ("Add", "AddV2") and b["name"] in u["inputs"]]
            assert adds, (L["name"], "the two convolutions are not summed")
            claimed.add(adds[0]["name"])
            tensor[L["dst"]] = follow_chain(adds[0]["name"], L)
            continue
        src = tensor[g[0]["src"]]
        if L["kind"] == "head_softmax":
            c = conv_of(src, 1, g[0]["channels"], cout, L["name"])
            last = follow_chain(c["name"], L)
            assert by[last]["op"] == "Softmax", (L["name"], last)
            continue
        c = conv_of(src, g[0]["ks"], g[0]["channels"], cout, L["name"])
        start = c["name"]
        if L["summed_shortcut_ks"]:
            k2 = L["summed_shortcut_ks"]
            b = conv_of(src, k2, g[0]["channels"], cout, L["name"] + " shortcut")
            adds = [u for u in users.get(c["name"], []) if u["op"] in ("Add", "AddV2") and b["name"] in u["inputs"]]
            assert adds, (L["name"], "main and shortcut convolutions are not summed")
            claimed.add(adds[0]["name"])
            start = adds[0]["name"]
        tensor[L["dst"]] = follow_chain(start, L)
    left = [o["name"] for o in ops if o["name"] not in claimed]
    assert not left, ("ops of the saved graph no launch stands for", left)


@pytest.mark.parametrize("name", sorted(CASES))
def test_engine_wiring_matches_the_saved_graph(name):
    hp = model.KNOWN_HP[CASES[name]]
    check_wiring(umx.describe_graph(hp), load(name))


def test_a_miswired_graph_is_caught():
    hp = model.KNOWN_HP["nucleiDAPI1-5"]
    good, fx = umx.describe_graph(hp), load("nucleiDAPI1-5")
    idx = {L["name"]: i for i, L in enumerate(good["launches"])}
    # swapped concat order
    bad = copy.deepcopy(good)
    bad["launches"][idx["lu1.conv"]]["groups"].reverse()
    with pytest.raises(AssertionError):
        check_wiring(bad, fx)
    # skip taken from the wrong level
    bad = copy.deepcopy(good)
    bad["launches"][idx["lu1.conv"]]["groups"][0]["src"] = good["launches"][idx["ld1.conv"]]["dst"]
    with pytest.raises(AssertionError):
        check_wiring(bad, fx)
    # pre-pool instead of pooled skip: the saved graph concatenates .../maxpool; a graph that concatenated the LeakyRelu output fails
    fx2 = copy.deepcopy(fx)
    for o in fx2["ops"]:
        if o["name"] == "upsampling/lu1/concat":
            o["inputs"][0] = "downsampling/ld0/LeakyRelu"
    with pytest.raises(AssertionError):
        check_wiring(good, fx2)
    # a layer that does not pool, a BatchNorm on the wrong side of the activation, another epsilon / slope
    for key, val in (("max_pool", 0), ("batch_norm", "after_activation"), ("activation", "relu")):
        bad = copy.deepcopy(good)
        bad["launches"][idx["ld0.conv"]][key] = val
        with pytest.raises(AssertionError):
            check_wiring(bad, fx)
    for key, val in (("bn_epsilon", 1e-5), ("leaky_slope", 0.01)):
        bad = copy.deepcopy(good)
        bad[key] = val
        with pytest.raises(AssertionError):
            check_wiring(bad, fx)


@pytest.mark.parametrize("name", ["nucleiDAPI1-5", "nucleiDAPILAMIN"])
def test_blob_tensor_shapes_match_the_checkpoint_index(name):
    """Every tensor the weight blob is cut into has the name and shape the reference's checkpoint index lists (solo and duo;
    their data shards are not shipped, the index is)."""
    hp = model.KNOWN_HP[name]
    shapes = load(name)["index_shapes"]
    for tname, shape in model.tensor_specs(hp):
        ck = model._ckpt_name(hp, tname)
        assert ck in shapes, (tname, ck)
        assert shapes[ck] == list(shape), (tname, shapes[ck], shape)
