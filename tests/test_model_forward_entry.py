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
