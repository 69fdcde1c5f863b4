"""Host logic of the product (no GPU): the launch plans describe the same networks as the oracle's
independent procedural statement, buffers assigned by liveness never alias a live tensor, and the
analytic work matches SURVEY §6."""
import pytest
import torch

import gvcnn_tf_amd as gv
from gvcnn_tf_amd import backbones
from oracle import backbone as OB


def _plan(name, nb, size):
    return backbones.make_plan(name, nb, size, size, torch.device("cpu"))


@pytest.mark.parametrize("name,size", [("inception_v3", 224), ("resnet_v2_50", 224)])
def test_param_names_and_shapes_match_oracle(name, size):
    p = _plan(name, 1, size)
    assert p.param_shapes() == OB.trace_param_shapes(name)


@pytest.mark.parametrize("name,size,gflop", [("inception_v3", 224, 5.672), ("inception_v3", 299, 11.422),
                                              ("resnet_v2_50", 224, 6.960)])
def test_flops_per_view_match_survey(name, size, gflop):
    p = _plan(name, 1, size)
    assert abs(p.total_flops / 1e9 - gflop) < 0.002 * gflop
    assert len(p.filters) == (94 if name == "inception_v3" else 53)          # slim.conv2d call sites
    n_launch = sum(1 for op in p.ops if op["kind"] == "conv")
    assert n_launch == (94 - 3 * 9 if name == "inception_v3" else 53)       # 9 blocks fuse 4 1x1s (3 siblings + the pooled branch)


def test_end_point_shapes():
    p = _plan("inception_v3", 2, 299)
    ep = p.end_points
    for k, (hw, c) in {"Conv2d_1a_3x3": (149, 32), "MaxPool_3a_3x3": (73, 64), "MaxPool_5a_3x3": (35, 192),
                       "Mixed_5b": (35, 256), "Mixed_5d": (35, 288), "Mixed_6a": (17, 768),
                       "Mixed_6e": (17, 768), "Mixed_7a": (8, 1280), "Mixed_7c": (8, 2048)}.items():
        assert (ep[k].nb, ep[k].h, ep[k].w, ep[k].c) == (2, hw, hw, c), k
    assert list(ep) == backbones.INCEPTION_ENDPOINTS
    p = _plan("resnet_v2_50", 2, 224)
    assert (p.end_points["resnet_v2_50/block3"].h, p.end_points["resnet_v2_50/block3"].c) == (7, 1024)
    assert (p.end_points["resnet_v2_50/block4"].h, p.end_points["resnet_v2_50/block4"].c) == (7, 2048)
    with pytest.raises(ValueError):
        backbones.build_inception_v3(backbones.BackbonePlan(1, 75, 75), final_endpoint="Mixed_8a")


@pytest.mark.parametrize("name,size", [("inception_v3", 75), ("resnet_v2_50", 64)])
def test_buffer_assignment_never_aliases_live_tensors(name, size):
    p = _plan(name, 2, size)
    vmap = p._phys
    # replay: a physical buffer written by op i must not hold a tensor that is read at or after i
    last_read = {}
    for i, op in enumerate(p.ops):
        for key in ("x", "res"):
            t = op.get(key)
            if t is not None and t.vbuf >= 0:
                last_read[t.vbuf] = i
    persistent = {v for v, vb in enumerate(p.vbufs) if vb[1]}
    owner = {}                       # phys -> vbuf currently stored
    for i, op in enumerate(p.ops):
        for key in ("y", "y2"):
            t = op.get(key)
            if t is None:
                continue
            ph = vmap[t.vbuf]
            prev = owner.get(ph)
            if prev is not None and prev != t.vbuf:
                assert prev not in persistent, (op["name"], "overwrites a kept end point")
                assert last_read.get(prev, -1) < i, (op["name"], "overwrites a live tensor")
            owner[ph] = t.vbuf
        for key in ("x", "res"):
            t = op.get(key)
            if t is not None and t.vbuf >= 0:
                assert owner.get(vmap[t.vbuf]) == t.vbuf, (op["name"], "reads a recycled buffer")
        # in/out of one op never share a physical buffer
        outs = {vmap[op[k].vbuf] for k in ("y", "y2") if op.get(k) is not None}
        ins = {vmap[op[k].vbuf] for k in ("x", "res") if op.get(k) is not None and op[k].vbuf >= 0}
        assert not (outs & ins), op["name"]
    # reuse actually happens
    assert len(set(vmap.values())) < len(p.vbufs) / 2


def test_concat_slices_cover_block_outputs():
    """Every channel of an Inception block output is written by exactly one producer (no tf.concat)."""
    p = _plan("inception_v3", 1, 75)
    cover = {}
    for op in p.ops:
        y = op["y"]
        cover.setdefault(y.vbuf, []).append((y.off % y.ld, y.off % y.ld + y.c, y.ld))
        if op.get("split"):                                  # fused siblings: the rest go to scratch
            y2 = op["y2"]
            assert op["cout"] == y.c + y2.c and op["split"] == y.c
    for name in ("Mixed_5b", "Mixed_6a", "Mixed_6e", "Mixed_7a", "Mixed_7c"):
        t = p.end_points[name]
        spans = sorted(cover[t.vbuf])
        assert spans[0][0] == 0 and spans[-1][1] == t.c
        for (a0, a1, _), (b0, b1, _) in zip(spans, spans[1:]):
            assert a1 == b0, (name, spans)


def test_resnet_fusions():
    """shortcut+residual is the conv3 epilogue; the next unit's preact is its second output."""
    p = _plan("resnet_v2_50", 1, 64)
    conv3 = [op for op in p.ops if op["name"].endswith("/conv3")]
    assert len(conv3) == 16 and all(op["res"] is not None for op in conv3)
    assert sum(op["y2"] is not None for op in conv3) == 15          # last unit has no successor
    assert sum(1 for op in p.ops if op["kind"] == "ssa") == 1       # only the very first preact
    assert sum(1 for op in p.ops if op["kind"] == "pool") == 1 + 3  # pool1 + 3 strided identity shortcuts


def test_head_param_names():
    H = gv.params.init_head_params(3, 16, 32, 5)
    assert sorted(H) == sorted(["dense/kernel", "dense/bias", "dense_1/kernel", "dense_1/bias",
                                "dense_2/kernel", "dense_2/bias", "dense_3/kernel", "dense_3/bias"])
    assert H["dense_3/kernel"].shape == (32, 5) and H["dense/kernel"].shape == (16, 1)


def test_sibling_fusion_is_optional_and_equivalent_in_work():
    a = backbones.BackbonePlan(1, 75, 75)
    backbones.build_inception_v3(a, fuse_siblings=False)
    b = backbones.BackbonePlan(1, 75, 75)
    backbones.build_inception_v3(b, fuse_siblings=True)
    assert sum(1 for op in a.ops if op["kind"] == "conv") == 94
    assert abs(a.total_flops - b.total_flops) < 1e-6 * a.total_flops
    assert a.param_shapes() == b.param_shapes()


def test_lane_schedule_orders_every_hazard():
    """Branch-level concurrency: replay the happens-before relation that stream order + the recorded
    cross-lane dependencies give, and check every conflicting pair of buffer accesses is ordered."""
    p = _plan("inception_v3", 2, 75)
    assert p.lanes_used == 3
    vmap = p._phys
    hb = []                                   # hb[i] = ops complete before op i may start
    last_on_lane = {}
    for i, op in enumerate(p.ops):
        s = set()
        prev = last_on_lane.get(op["lane"])
        if prev is not None:
            s |= hb[prev] | {prev}
        for d in op.get("deps", ()):
            assert d < i and p.ops[d]["lane"] != op["lane"]
            s |= hb[d] | {d}
        hb.append(s)
        last_on_lane[op["lane"]] = i

    def accesses(op):
        out = []
        for key, w in (("x", False), ("res", False), ("y", True), ("y2", True)):
            t = op.get(key)
            if t is not None and t.vbuf >= 0:
                lo = t.off % t.ld
                out.append((vmap[t.vbuf], w, t.vbuf, lo, lo + t.c))
        return out

    n_cross = 0
    for i, oi in enumerate(p.ops):
        for j in range(i):
            oj = p.ops[j]
            for (pi, wi, vi, li, hi_) in accesses(oi):
                for (pj, wj, vj, lj, hj) in accesses(oj):
                    if pi != pj or not (wi or wj):
                        continue
                    if vi == vj and (hi_ <= lj or hj <= li):
                        continue
                    assert j in hb[i], (oi["name"], "is not ordered after", oj["name"])
                    n_cross += oi["lane"] != oj["lane"]
    assert n_cross > 20                       # the check is not vacuous
    # independent branches really are unordered (that is the point)
    names = {op["name"]: k for k, op in enumerate(p.ops)}
    a = names["InceptionV3/Mixed_6b/Branch_1/Conv2d_0b_1x7"]
    b = names["InceptionV3/Mixed_6b/Branch_2/Conv2d_0b_7x1"]
    assert a not in hb[b] and b not in hb[a]
    # single-lane plans carry no schedule
    q = backbones.make_plan("inception_v3", 1, 75, 75, torch.device("cpu"), lanes=False)
    assert all(not op.get("deps") for op in q.ops)


def test_resnet_16bit_plan_defers_the_preactivation_of_identity_units():
    """16-bit storage: conv3 of a unit followed by an identity unit writes the sum only, and that unit's conv1 carries
    the folded pre-activation BatchNorm for its loader (the 10 identity boundaries of block1-block3; the block boundaries,
    where the projection shortcut reads the pre-activation too, and block4 keep the stored second output).  fp32 storage: the stored form everywhere."""
    p = backbones.make_plan("resnet_v2_50", 2, 64, 64, torch.device("cpu"), dtype="bf16", fuse_chain=False)
    convs = [op for op in p.ops if op["kind"] == "conv"]
    pre = [op for op in convs if op.get("xpre") is not None]
    assert len(pre) == 10 and all(op["name"].endswith("/conv1") and op["kh"] == 1 and op["pad_t"] == 0 for op in pre)
    assert sum(1 for op in convs if op["y2"] is not None) == 5
    for op in pre:                                   # the input of such a conv1 is the previous unit's sum itself
        prev = convs[convs.index(op) - 1]
        assert prev["name"].endswith("/conv3") and prev["y"] is op["x"] and prev["y2"] is None
        assert op["xpre"] == (prev["scale2_off"], prev["shift2_off"])
    q = backbones.make_plan("resnet_v2_50", 2, 64, 64, torch.device("cpu"))
    assert all(op.get("xpre") is None for op in q.ops) and sum(1 for op in q.ops if op.get("y2") is not None) == 15
    assert p.param_shapes() == q.param_shapes()


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
def test_16bit_plans_issue_conv_and_max_pool_as_one_launch(dtype):
    """GV_CONV_MAXPOOL3S2 / _SAME at plan level (host logic only): the 16-bit plans replace `Conv2d_2b_3x3 -> MaxPool_3a_3x3`
    (nets/inception_v3.py:111-113) and `conv1 -> pool1` (nets/resnet_v2.py:178-181) by one op whose destination is the pooled
    tensor; the arithmetic work is unchanged; a tapped producer, an odd conv1 map (SAME pads (1, 1) there) and fp32 storage
    keep the two launches."""
    cpu = torch.device("cpu")
    p = backbones.make_plan("inception_v3", 2, 224, 224, cpu, dtype=dtype)
    names = [op["name"] for op in p.ops]
    assert "MaxPool_3a_3x3" not in names and "Conv2d_2b_3x3" not in p.end_points
    op = next(o for o in p.ops if o["name"].endswith("Conv2d_2b_3x3"))
    assert op["maxpool"] == "VALID" and (op["oh"], op["ow"]) == (109, 109) and (op["y"].h, op["y"].w, op["y"].c) == (54, 54, 64)
    assert p.end_points["MaxPool_3a_3x3"] is op["y"] or p.end_points["MaxPool_3a_3x3"].vbuf == op["y"].vbuf
    assert abs(p.total_flops / 2 / 1e9 - 5.672) < 0.002 * 5.672
    assert sum(1 for o in p.ops if o["kind"] == "pool") == 13 - 1                         # 4 max + 9 average pools before
    tapped = backbones.make_plan("inception_v3", 2, 224, 224, cpu, dtype=dtype, raw_tap="Conv2d_2b_3x3")
    assert "MaxPool_3a_3x3" in [o["name"] for o in tapped.ops] and "Conv2d_2b_3x3" in tapped.end_points
    off = backbones.make_plan("inception_v3", 2, 224, 224, cpu, dtype=dtype, fuse_maxpool=False)
    assert "MaxPool_3a_3x3" in [o["name"] for o in off.ops]
    assert off.param_shapes() == p.param_shapes()

    r = backbones.make_plan("resnet_v2_50", 2, 224, 224, cpu, dtype=dtype)
    op = next(o for o in r.ops if o["name"] == "resnet_v2_50/conv1")
    assert op["maxpool"] == "SAME" and (op["oh"], op["ow"]) == (112, 112) and (op["y"].h, op["y"].w) == (56, 56)
    assert not any(o["name"].endswith("/pool1") for o in r.ops)
    assert abs(r.total_flops / 2 / 1e9 - 6.960) < 0.002 * 6.960
    odd = backbones.make_plan("resnet_v2_50", 2, 97, 97, cpu, dtype=dtype)                # conv1: 49 x 49
    assert any(o["name"].endswith("/pool1") for o in odd.ops)
    for name in ("inception_v3", "resnet_v2_50"):
        f32 = backbones.make_plan(name, 2, 224, 224, cpu)
        assert not any(o.get("maxpool") for o in f32.ops if o["kind"] == "conv")
    # fp32 storage under the three-plane math (the bench line): the Inception pair is one launch where the halo kernel runs
    # its 30-pixel strip form (a Conv2d_2b map wider than 96 pixels), two launches on smaller maps; ResNet keeps two
    x3 = backbones.make_plan("inception_v3", 2, 224, 224, cpu, math="bf16x3")
    op = next(o for o in x3.ops if o["name"].endswith("Conv2d_2b_3x3"))
    assert op["maxpool"] == "VALID" and (op["y"].h, op["y"].w, op["y"].c) == (54, 54, 64) and not op["y"].p3
    assert "MaxPool_3a_3x3" not in [o["name"] for o in x3.ops]
    assert abs(x3.total_flops / 2 / 1e9 - 5.672) < 0.002 * 5.672
    small = backbones.make_plan("inception_v3", 2, 128, 128, cpu, math="bf16x3")             # Conv2d_2b: 61 x 61
    assert "MaxPool_3a_3x3" in [o["name"] for o in small.ops]
    off3 = backbones.make_plan("inception_v3", 2, 224, 224, cpu, math="bf16x3", fuse_maxpool=False)
    assert "MaxPool_3a_3x3" in [o["name"] for o in off3.ops] and off3.param_shapes() == x3.param_shapes()
    r3 = backbones.make_plan("resnet_v2_50", 2, 224, 224, cpu, math="bf16x3")
    assert not any(o.get("maxpool") for o in r3.ops if o["kind"] == "conv")


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
def test_16bit_resnet_plan_chains_conv3_into_the_next_conv1(dtype):
    """gv_bottleneck_chain_fwd / gv_bottleneck_unit_fwd at plan level (host logic only): inside blocks 1 and 2 (bottleneck
    depth 64 / 128) conv3 of a unit, the next unit's pre-activation and its conv1 (nets/resnet_v2.py:87-91, :75, :83-84) are
    ONE op — 2 + 3 of them — and by default the unit's conv2 (:85-86) runs in front of it in the same launch: ten launches
    fewer (fuse_unit=False: five).  The deeper blocks keep the deferred pre-activation; same variables, same arithmetic work,
    same end points; fp32 storage and fuse_chain=False keep the separate launches."""
    cpu = torch.device("cpu")
    off = backbones.make_plan("resnet_v2_50", 2, 224, 224, cpu, dtype=dtype, fuse_chain=False)
    assert not any(op.get("chain") for op in off.ops)
    pairs = [op for op in backbones.make_plan("resnet_v2_50", 2, 224, 224, cpu, dtype=dtype).ops if op.get("split")]
    # (a depth-changing unit's conv1 + projection shortcut also run as one GEMM in blocks 2 - 4: three launches fewer, resnet_unit1_pair)
    assert [(op["split"], op["cout"], op["relu_cols"]) for op in pairs] == [(128, 640, 128), (256, 1280, 256), (512, 2560, 512)]   # (block1's: slower, not fused)
    assert all(op["name"].split("+")[0].endswith("unit_1/bottleneck_v2/conv1") and op["name"].endswith("unit_1/bottleneck_v2/shortcut")
               for op in pairs)
    # The FIRST unit changes the depth: its projection shortcut rides inside conv3's GEMM (GV_CHAIN_PROJ: x = [conv2 | preact],
    # 128 channels, no shortcut operand, no shortcut launch), its conv2 stays a launch of its own
    for fuse_unit, fewer, fronts in (("all", 13, 4), (True, 10, 1), (False, 9, 0)):       # (default: whole units at d = 64 only)
        p = backbones.make_plan("resnet_v2_50", 2, 224, 224, cpu, dtype=dtype, fuse_unit=fuse_unit)
        chains = [op for op in p.ops if op.get("chain")]
        assert [op["x"].c for op in chains] == [128, 64, 128, 128, 128]
        pj = chains[0]
        assert pj["chain"]["proj"] and pj["res"] is None and pj["y"].c == 256 and pj["y2"].c == 64 and not pj["chain"].get("front")
        assert pj["name"].split("+")[0].endswith("block1/unit_1/bottleneck_v2/shortcut")
        assert not any(o["name"].endswith("block1/unit_1/bottleneck_v2/shortcut") for o in p.ops)     # no shortcut launch
        c2 = next(o for o in p.ops if o["name"].endswith("block1/unit_1/bottleneck_v2/conv2"))
        c1 = next(o for o in p.ops if o["name"].endswith("block1/unit_1/bottleneck_v2/conv1"))
        assert c2["y"].vbuf == pj["x"].vbuf and c2["y"].off == pj["x"].off and c2["y"].ld == 128       # conv2 writes channels [0, 64)
        assert c1["x"].vbuf == pj["x"].vbuf and c1["x"].off == pj["x"].off + 64 and p.ops[0]["y"].off == c1["x"].off   # preact: [64, 128)
        chains = chains[1:]
        assert sum(1 for op in chains if op["chain"].get("front")) == fronts
        assert all(bool(op["chain"].get("front")) == (fuse_unit == "all" or (fuse_unit is True and op["x"].c == 64)) for op in chains)
        assert len(p.ops) == len(off.ops) - fewer
        for op in chains:
            parts = op["name"].split("+")
            a, b = parts[-2], parts[-1]
            assert a.endswith("/conv3") and b.endswith("/conv1") and a.split("/")[1] == b.split("/")[1]      # same block
            assert int(b.split("/unit_")[1].split("/")[0]) == int(a.split("/unit_")[1].split("/")[0]) + 1      # consecutive units
            assert op["y"].c == 4 * op["x"].c and op["y2"].c == op["x"].c and op["res"].c == op["y"].c
            nxt = next(o for o in p.ops if o["name"].split("+")[0] == b.replace("/conv1", "/conv2"))
            assert nxt["x"] is op["y2"] and p.ops.index(nxt) > p.ops.index(op)                                # conv2 reads the chain's z
            assert not any(o["name"] == b for o in p.ops)                                                     # no separate conv1
            own_conv2 = a.replace("/conv3", "/conv2")
            if op["chain"].get("front"):                                                                      # ... nor conv2: it is in front
                assert parts[0] == own_conv2 and len(parts) == 3 and not any(o["name"] == own_conv2 for o in p.ops)
                prev = next(o for o in p.ops if a.replace("/conv3", "/conv1") in o["name"].split("+"))
                assert op["x"] is (prev["y2"] if prev.get("chain") else prev["y"])                            # x is this unit's conv1
            else:
                assert len(parts) == 2 and any(o["name"] == own_conv2 and o["y"] is op["x"] for o in p.ops)
        pre = [op for op in p.ops if op["kind"] == "conv" and op.get("xpre") is not None]
        assert len(pre) == 5 and all(op["x"].c == 1024 for op in pre)                                         # block3's identity units
        assert p.param_shapes() == off.param_shapes()
        assert abs(p.total_flops - off.total_flops) < 1e-6 * off.total_flops
        assert set(p.end_points) == set(off.end_points)
        # the first unit's pre-activation rides on the fused conv1 -> pool1 launch (GV_CONV_POOL_ACT2): no stand-alone pass
        assert sum(1 for op in p.ops if op["kind"] == "ssa") == 0 and p.ops[0].get("pool_act") and p.ops[0]["maxpool"] == "SAME"
    two = backbones.make_plan("resnet_v2_50", 2, 224, 224, cpu, dtype=dtype, fuse_maxpool=False)
    assert sum(1 for op in two.ops if op["kind"] == "ssa") == 1 and not any(op.get("pool_act") for op in two.ops)
    assert len(two.ops) == len(off.ops) - 10 + 2 and two.param_shapes() == off.param_shapes()
    ssa = next(op for op in two.ops if op["kind"] == "ssa")                       # (the stand-alone pre-activation writes the slice too)
    pj = next(op for op in two.ops if op.get("chain") and op["chain"].get("proj"))
    assert ssa["y"].vbuf == pj["x"].vbuf and ssa["y"].off == pj["x"].off + 64
    f32 = backbones.make_plan("resnet_v2_50", 2, 224, 224, cpu)
    assert not any(op.get("chain") for op in f32.ops)
