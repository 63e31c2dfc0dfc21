"""The streaming aggregation (gnnome_node_aggregate_stream_f32, gated_gcn_full.py:111-137 with every e row read once): its schedule
against a plain-Python restatement, bit for bit; its result against the fp64 statement of the node update's contract and against the
one-wave-per-node kernel; pending nodes, far rows, duplicates, isolated nodes, slot overflow, every built width."""
import numpy as np
import pytest
import torch

import cpu_ops
from gnnome_amd.synth import make_graph

pytestmark = pytest.mark.gpu

DEV = torch.device("cuda", 0)


def _views(src, dst, n):
    from gnnome_amd import ops
    return ops.GraphViews(torch.as_tensor(src, dtype=torch.int32, device=DEV), torch.as_tensor(dst, dtype=torch.int32, device=DEV), n)


def _graph(kind, n, e, seed=1):
    if kind == "hand":
        # isolated node 5, a sink, a source, self-loops, a duplicate pair, a small hub
        src = [0, 1, 2, 2, 3, 3, 4, 0, 0, 6, 7, 7, 7, 7, 1, 1]
        dst = [1, 2, 2, 3, 4, 4, 0, 2, 3, 7, 6, 0, 1, 2, 7, 7]
        return np.array(src), np.array(dst), 9
    g = make_graph(n, e, seed, kind)
    return g["src"].numpy(), g["dst"].numpy(), g["num_nodes"]


def _schedule(views, chunks, slots=62):
    from gnnome_amd import ops
    return ops.StreamSchedule(views, chunks=chunks, slots=slots)


@pytest.mark.parametrize("kind,n,e,chunks,slots", [("hand", 0, 0, 1, 62), ("hand", 0, 0, 3, 2), ("banded", 2000, 20000, 7, 62),
                                                    ("banded", 2000, 20000, 16, 12), ("uniform", 600, 5000, 4, 62),
                                                    ("banded", 20000, 200000, 64, 40)])
def test_schedule_equals_its_python_restatement(kind, n, e, chunks, slots):
    src, dst, n = _graph(kind, n, e)
    views = _views(src, dst, n)
    sched = _schedule(views, chunks, slots)
    want = cpu_ops.stream_schedule(views, chunks, 16, slots)
    assert np.array_equal(sched.chunk_node.cpu().numpy(), want["chunk_node"])
    assert np.array_equal(sched.chunk_steps.cpu().numpy(), want["chunk_steps"])
    assert np.array_equal(sched.edge_meta.cpu().numpy().astype(np.int64), want["edge_meta"])
    node_pend = sched.node_pend.cpu().numpy()
    assert set(np.flatnonzero(node_pend >= 0).tolist()) == want["pending"]
    assert sched.num_pending == len(want["pending"]) and sched.num_far == want["far"] and sched.num_overflow == want["overflow"]
    pend_nodes = sched.pend_nodes.cpu().numpy()[:sched.num_pending]
    assert sorted(pend_nodes.tolist()) == sorted(want["pending"]) and all(node_pend[s] == i for i, s in enumerate(pend_nodes))
    steps = sched.steps.cpu().numpy()
    in_ptr = views.in_ptr.cpu().numpy()
    for c in range(chunks):
        n0 = int(want["chunk_node"][c])
        base = int(in_ptr[n0]) // 16 + n0
        got = steps[base:base + int(want["chunk_steps"][c])]
        for row, (p0, node, word, pend) in zip(got, want["steps"][c]):
            assert (int(row[0]), int(row[1]), int(row[2])) == (p0, node, word) and (int(row[3]) >= 0) == pend
            assert int(row[3]) == int(node_pend[node])


def _inputs(views, H, seed=0, ld=None):
    g = torch.Generator().manual_seed(seed)
    n, e = views.num_nodes, views.num_edges
    ld = 5 * H if ld is None else ld
    P = torch.randn(n, ld, generator=g)
    return {"e": (torch.randn(e, H, generator=g) * 2).to(DEV), "P": P.to(DEV), "h": torch.randn(n, H, generator=g).to(DEV),
            "scale": (1 + 0.1 * torch.randn(H, generator=g)).to(DEV), "shift": (0.1 * torch.randn(H, generator=g)).to(DEV)}


def _run(views, t, H, sched):
    """The C-ABI entry, called directly with an explicit schedule."""
    from gnnome_amd import _lib, ops
    lib = _lib.load()
    A1, A2, A3 = (t["P"][:, i * H:(i + 1) * H] for i in range(3))
    out = torch.full((views.num_nodes, H), float("nan"), device=DEV)
    pend = torch.empty((max(sched.num_pending, 1), 3, H), device=DEV)
    p = ops._ptr
    _lib.check(lib.gnnome_node_aggregate_stream_f32(p(t["e"]), H, views.num_nodes, views.num_edges, p(A1), p(A2), p(A3), t["P"].stride(0), p(views.in_ptr), p(views.srt_src),
                                                    p(views.out_ptr), p(views.out_pos), p(views.out_dst), p(t["h"]), H, p(out), p(t["scale"]),
                                                    p(t["shift"]), sched.chunks, 16, 62, p(sched.chunk_node), p(sched.chunk_steps), p(sched.steps),
                                                    p(sched.edge_meta), p(sched.node_pend), p(sched.pend_nodes), p(sched.counters), sched.num_pending,
                                                    p(pend), ops._stream(DEV)), "node_aggregate_stream_f32")
    torch.cuda.synchronize()
    return out


def _contract(views, t, H):
    cv = cpu_ops.CpuViews(views.srt_src.cpu(), views.srt_dst.cpu(), views.num_nodes)   # (sorted order in = sorted order out)
    P = t["P"].cpu().double()
    A1, A2, A3 = (P[:, i * H:(i + 1) * H] for i in range(3))
    return cpu_ops.node_aggregate(t["e"].cpu().double(), A1, A2, A3, cv, t["h"].cpu().double(), 0, t["scale"].cpu().double(), t["shift"].cpu().double())


@pytest.mark.parametrize("kind,n,e,chunks,slots", [("hand", 0, 0, 1, 62), ("hand", 0, 0, 3, 2), ("banded", 2000, 20000, 7, 62),
                                                    ("banded", 2000, 20000, 16, 12), ("uniform", 600, 5000, 4, 62),
                                                    ("banded", 20000, 200000, 64, 62), ("banded", 20000, 200000, 300, 30)])
@pytest.mark.parametrize("H", [64, 128, 256])
def test_stream_aggregate_against_the_contract(kind, n, e, chunks, slots, H):
    src, dst, n = _graph(kind, n, e)
    views = _views(src, dst, n)
    sched = _schedule(views, chunks, slots)
    t = _inputs(views, H)
    got = _run(views, t, H, sched)
    want = _contract(views, t, H)
    scale = want.abs().max().item()
    assert torch.isfinite(got).all()
    assert (got.cpu().double() - want).abs().max().item() < 1e-5 * max(scale, 1.0)
    again = _run(views, t, H, sched)
    assert torch.equal(got, again)   # bit-reproducible


def test_stream_and_gather_kernels_agree_through_ops_and_the_policy_picks():
    from gnnome_amd import ops
    g = make_graph(60000, 600000, 1, "banded")
    views = _views(g["src"].numpy(), g["dst"].numpy(), g["num_nodes"])
    H = 128
    t = _inputs(views, H)
    A1, A2, A3 = (t["P"][:, i * H:(i + 1) * H] for i in range(3))
    sched = views.stream_schedule()
    assert sched.usable and sched.far_fraction < 0.2, (sched.why, sched.far_fraction)
    saved = ops.STREAM_AGGREGATE
    try:
        ops.STREAM_AGGREGATE = "auto"
        got = ops.node_aggregate(t["e"], A1, A2, A3, views, t["h"], 0, t["scale"], t["shift"])
        ops.STREAM_AGGREGATE = False
        old = ops.node_aggregate(t["e"], A1, A2, A3, views, t["h"], 0, t["scale"], t["shift"])
    finally:
        ops.STREAM_AGGREGATE = saved
    assert not torch.equal(got, old)   # (another association: the streaming kernel really ran)
    assert (got - old).abs().max().item() < 1e-5 * old.abs().max().item()
    # reversed views share the schedule; a uniform graph is left to the gathering kernel
    assert views.reversed().stream_schedule() is sched
    gu = make_graph(60000, 600000, 1, "uniform")
    vu = _views(gu["src"].numpy(), gu["dst"].numpy(), gu["num_nodes"])
    assert not vu.stream_schedule().usable and "far" in vu.stream_schedule().why
