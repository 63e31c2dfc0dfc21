"""gnnome_model_forward_f32 (csrc/model_forward.cpp, round 6): models/full_graph.py:22-30 - what inference.py:440 calls once per graph - as ONE
call into the library.  It sequences the per-kernel entries exactly as gnnome_amd/engine.py::run_stack does, so the bar is bits: every case below
runs the model twice, call by call (engine.ONE_CALL_FORWARD = False) and through the one entry, and compares with torch.equal; the goldens and
the oracle pin the values themselves in tests/test_hip_parity.py, which runs through the one entry by default."""
import pytest
import torch

import gnnome_amd
from gnnome_amd import engine, ops
from gnnome_amd.graph import views_for
from gnnome_amd.synth import make_graph, random_state_dict

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda", 0)


def _model(hidden, hs, norm, layers=3, seed=1, arithmetic="auto"):
    m = gnnome_amd.SymGatedGCNModel(2, 2, hidden, 16, layers, hs, norm).eval()
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for p in m.parameters():
            p.copy_(torch.randn(p.shape, generator=g) * (0.5 / max(p.shape[-1], 1) ** 0.5))
        for mod in m.modules():
            if isinstance(mod, torch.nn.BatchNorm1d):
                mod.running_mean.copy_(torch.randn(mod.running_mean.shape, generator=g) * 0.3)
                mod.running_var.copy_(torch.rand(mod.running_var.shape, generator=g) + 0.2)
    m.arithmetic = arithmetic
    return m.to(dev())


def _both_ways(model, graph, x, e):
    old = engine.ONE_CALL_FORWARD
    try:
        engine.ONE_CALL_FORWARD = False
        a = model(graph, x, e)
        engine.ONE_CALL_FORWARD = True
        b = model(graph, x, e)
    finally:
        engine.ONE_CALL_FORWARD = old
    torch.cuda.synchronize()
    return a, b


@pytest.mark.parametrize("hidden,hs,norm,arith", [(64, 64, "batch", "auto"), (128, 64, "batch", "auto"), (256, 64, "batch", "auto"), (128, 128, "batch", "fast"),
                                                  (256, 128, "layer", "auto"), (128, 32, "layer", "auto"), (128, 64, "batch", "reference"),
                                                  (256, 64, "batch", "reference"), (96, 48, "batch", "auto"), (160, 128, "layer", "auto")])
def test_one_call_equals_the_call_by_call_sequence(hidden, hs, norm, arith):
    n, e = 3000, 31000
    g = make_graph(n, e, seed=hidden + hs)
    model = _model(hidden, hs, norm, arithmetic=arith)
    x = torch.randn(n, 2, generator=torch.Generator().manual_seed(2)).to(dev())
    graph = (g["src"], g["dst"], n)
    a, b = _both_ways(model, graph, x, g["e"].to(dev()))
    assert a.shape == (e, 1) and torch.isfinite(a).all()
    assert torch.equal(a, b)


def test_one_call_on_reversed_and_renumbered_views_and_cpu_inputs():
    n, e = 2500, 26000
    g = make_graph(n, e, seed=5)
    model = _model(128, 64, "batch")
    views = views_for((g["src"], g["dst"], n), dev())
    x = torch.randn(n, 2, generator=torch.Generator().manual_seed(3))
    for vw in (views, views.reversed()):
        a, b = _both_ways(model, vw, x.to(dev()), g["e"].to(dev()))
        assert torch.equal(a, b)
    perm = torch.randperm(n, generator=torch.Generator().manual_seed(4))
    renum = ops.GraphViews(g["src"].to(dev()), g["dst"].to(dev()), n, node_perm=perm)
    a, b = _both_ways(model, renum, x.to(dev()), g["e"].to(dev()))
    assert torch.equal(a, b)
    plain = model(views, x.to(dev()), g["e"].to(dev()))
    assert (torch.sigmoid(a) - torch.sigmoid(plain)).abs().max() < 1e-5   # renumbering reorders sums, nothing else
    a, b = _both_ways(model, (g["src"], g["dst"], n), x, g["e"])         # inference.py:388: everything on the CPU
    assert a.device.type == "cpu" and torch.equal(a, b)


def test_one_call_edge_cases():
    model = _model(128, 64, "batch", layers=2)
    # no edges, a single edge, isolated nodes only
    for n, src, dst in ((5, [], []), (2, [0], [1]), (64, [3, 3, 7], [7, 3, 3])):
        x = torch.randn(n, 2).to(dev())
        e = torch.randn(len(src), 2).to(dev())
        graph = (torch.tensor(src, dtype=torch.int32), torch.tensor(dst, dtype=torch.int32), n)
        a, b = _both_ways(model, graph, x, e)
        assert a.shape == (len(src), 1) and torch.equal(a, b)
    # no layers at all: encoders -> scorer
    bare = _model(128, 64, "batch", layers=0)
    g = make_graph(500, 4000, seed=9)
    a, b = _both_ways(bare, (g["src"], g["dst"], 500), torch.randn(500, 2).to(dev()), g["e"].to(dev()))
    assert torch.equal(a, b)


def test_one_call_follows_the_bf16x6_switch_and_the_range_fallback():
    n, e = 2000, 20000
    g = make_graph(n, e, seed=11)
    model = _model(128, 64, "batch")
    x = torch.randn(n, 2).to(dev())
    ed = g["e"].to(dev())
    with ops.bf16x6_arithmetic():
        a, b = _both_ways(model, (g["src"], g["dst"], n), x, ed)
    assert torch.equal(a, b)
    fp16 = model((g["src"], g["dst"], n), x, ed)
    assert not torch.equal(a, fp16) and (torch.sigmoid(a) - torch.sigmoid(fp16)).abs().max() < 1e-4



def test_separate_buffers_and_one_block_give_the_same_bits():
    """gnnome_model_forward_buffers_f32 (the default since round 6: buffers allocated one by one) against gnnome_model_forward_f32 on one workspace block."""
    old = ops.FORWARD_BUFFERS
    try:
        for hidden, hs in ((128, 64), (256, 64), (64, 32)):
            n, e = 4000, 41000
            g = make_graph(n, e, seed=hidden)
            model = _model(hidden, hs, "batch")
            x = torch.randn(n, 2, generator=torch.Generator().manual_seed(5)).to(dev())
            views = views_for((g["src"], g["dst"], n), dev())
            ops.FORWARD_BUFFERS = "separate"
            a = model(views, x, g["e"].to(dev())).clone()
            ops.FORWARD_BUFFERS = "block"
            b = model(views, x, g["e"].to(dev()))
            assert torch.equal(a, b)
        ops.FORWARD_BUFFERS = "separate"
        empty = model((torch.zeros(0, dtype=torch.int32), torch.zeros(0, dtype=torch.int32), 6), torch.randn(6, 2).to(dev()), torch.zeros(0, 2).to(dev()))
        assert empty.shape == (0, 1)
    finally:
        ops.FORWARD_BUFFERS = old


def test_placed_buffers_keep_the_bits_and_are_kept_per_stream():
    """ops._placed_buffers (round 6): from its third use on, a forward whose buffers' placement in HBM varies (each below 1 GiB, 64 MiB in all) runs in the
    fastest of a few candidate allocations per buffer group - same logits before, while and after; one kept set per (device, stream, shapes); nothing is kept
    under TUNE_PLACEMENT = False, for small graphs, or inside a captured forward."""
    from gnnome_amd.capture import CapturedForward
    n, e = 30000, 300000     # H = 128: 276 MB of buffers
    g = make_graph(n, e, seed=7)
    model = _model(128, 64, "batch")
    x = torch.randn(n, 2, generator=torch.Generator().manual_seed(5)).to(dev())
    ed = g["e"].to(dev())
    views = views_for((g["src"], g["dst"], n), dev())
    old = ops.TUNE_PLACEMENT, dict(ops._PLACED)
    try:
        ops._PLACED.clear()
        ops.TUNE_PLACEMENT = True
        outs = [model(views, x, ed).clone() for _ in range(ops.PLACEMENT_AFTER_USES + 2)]
        assert all(torch.equal(o, outs[0]) for o in outs)
        placed = [st for st in ops._PLACED.values() if st.bufs is not None]
        assert len(placed) == 1 and placed[0].tried and len(placed[0].log) == 1 + 3 * ops.PLACEMENT_CANDIDATES
        before = {k: t.data_ptr() for k, t in placed[0].bufs.items()}
        assert len(set(before.values())) == len(before)                                              # six (five) distinct buffers
        assert torch.equal(model(views, x, ed), outs[0]) and {k: t.data_ptr() for k, t in placed[0].bufs.items()} == before
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(ops.PLACEMENT_AFTER_USES + 1):
                o = model(views, x, ed)
        torch.cuda.current_stream().wait_stream(side)
        assert torch.equal(o, outs[0])
        sets = [st.bufs for st in ops._PLACED.values() if st.bufs is not None]
        assert len(sets) == 2 and not ({t.data_ptr() for t in sets[0].values()} & {t.data_ptr() for t in sets[1].values()})   # one set per stream
        cap = CapturedForward(model, views, x, ed)                                                    # a hipGraph keeps its own pool's buffers
        assert torch.equal(cap(), outs[0]) and len([st for st in ops._PLACED.values() if st.bufs is not None]) == 2
        ops._PLACED.clear()
        ops.TUNE_PLACEMENT = False
        for _ in range(ops.PLACEMENT_AFTER_USES + 2):
            assert torch.equal(model(views, x, ed), outs[0])
        assert not ops._PLACED
        ops.TUNE_PLACEMENT = True
        small = make_graph(2000, 20000, seed=8)                                                       # 18 MB of buffers: below the floor
        for _ in range(ops.PLACEMENT_AFTER_USES + 2):
            model((small["src"], small["dst"], 2000), torch.randn(2000, 2).to(dev()), small["e"].to(dev()))
        assert not ops._PLACED
    finally:
        ops.TUNE_PLACEMENT = old[0]
        ops._PLACED.clear()
        ops._PLACED.update(old[1])
