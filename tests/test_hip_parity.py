"""GPU parity tests: every C-ABI kernel against the checker backend, the whole model against the
reference goldens and the oracle, and size-independent properties at BASELINE config sizes.

Tolerances: kernels are compared with an fp64 evaluation of their contract (abs+rel 1e-5 scaled by
the operand magnitude - exact-fp32 accumulation error); model outputs use north_star's bar,
max |sigmoid(logit) - sigmoid(reference logit)| < 1e-4.
"""
import pytest
import torch

import cpu_ops
import gnnome_amd
from conftest import load_golden
from gnnome_amd import ops
from gnnome_amd.graph import reverse, views_for
from gnnome_amd.synth import make_graph, random_state_dict
from oracle.symgated_oracle import degree_features, model_from_state_dict

pytestmark = pytest.mark.gpu
PROB_TOL = 1e-4


def dev():
    return torch.device("cuda", 0)


def _rand_graph(n, e, seed, isolated=True):
    g = torch.Generator().manual_seed(seed)
    hi = n - 3 if isolated else n  # the last nodes stay isolated
    src = torch.randint(0, hi, (e,), generator=g).int()
    dst = torch.randint(0, hi, (e,), generator=g).int()
    return src, dst


def _assert_close(got, want64, scale=None, tol=1e-5):
    got = got.double().cpu()
    scale = want64.abs().max().item() if scale is None else scale
    err = (got - want64).abs().max().item()
    assert err <= tol * max(scale, 1.0), f"max abs err {err:.3e} vs scale {scale:.3e}"


def _views_pair(src, dst, n):
    return ops.GraphViews(src.to(dev()), dst.to(dev()), n), cpu_ops.CpuViews(src, dst, n)


# ------------------------------------------------------------------------------------ graph views

@pytest.mark.parametrize("n,e", [(8, 14), (1000, 20000), (5000, 3), (64, 0), (70000, 300000)])
def test_graph_views_bit_exact(n, e):
    src, dst = _rand_graph(n, e, seed=n + e, isolated=n > 8)
    gv, cv = _views_pair(src, dst, n)
    torch.cuda.synchronize()
    for name in ("in_ptr", "srt_src", "srt_dst", "srt_eid", "out_ptr", "out_pos", "out_dst"):
        assert torch.equal(getattr(gv, name).cpu(), getattr(cv, name)), name


def test_graph_views_rejects_out_of_range():
    with pytest.raises(IndexError):
        ops.GraphViews(torch.tensor([0, 9], dtype=torch.int32, device=dev()), torch.tensor([1, 2], dtype=torch.int32, device=dev()), 4)
    # a graph handed to the model is checked lazily (no host sync before the kernels are enqueued; endpoints clamped so that
    # nothing can fault) and the call still raises
    lazy = ops.GraphViews(torch.tensor([0, 9], dtype=torch.int32, device=dev()), torch.tensor([1, -2], dtype=torch.int32, device=dev()), 4,
                          validate="lazy")
    assert int(lazy.srt_src.max()) <= 3 and int(lazy.srt_dst.min()) >= 0
    with pytest.raises(IndexError):
        lazy.check_range()
    m = gnnome_amd.SymGatedGCNModel(2, 2, 64, 16, 2, 64, "batch").eval().to(dev())
    with pytest.raises(IndexError):
        m((torch.tensor([0, 9]), torch.tensor([1, 2]), 4), torch.randn(4, 2, device=dev()), torch.randn(2, 2, device=dev()))
    with pytest.raises(IndexError):   # sticky: a second look at the same views raises again
        lazy.check_range()

    class Graph:   # a graph OBJECT is cached per identity: the refused views must not be served on the second call
        def edges(self):
            return torch.tensor([0, 9]), torch.tensor([1, 2])

        def num_nodes(self):
            return 4

        def num_edges(self):
            return 2

    g = Graph()
    x, e = torch.randn(4, 2, device=dev()), torch.randn(2, 2, device=dev())
    for _ in range(2):
        with pytest.raises(IndexError):
            m(g, x, e)
    # every entry point that builds views from a caller's graph checks them: layer-level API, scorer, degree features
    from gnnome_amd import features
    h, eh = torch.randn(4, 64, device=dev()), torch.randn(2, 64, device=dev())
    with torch.no_grad():
        for call in (lambda: m.gnn.convs[0](g, h, eh), lambda: m.predictor(g, h, eh), lambda: features.degree_features(Graph()),
                     lambda: m.gnn.convs[0]((torch.tensor([0, 9]), torch.tensor([1, 2]), 4), h, eh)):
            with pytest.raises(IndexError):
                call()


# ------------------------------------------------------------------------------------ kernels

@pytest.mark.parametrize("hidden", [64, 128, 256])
def test_encode(hidden):
    g = torch.Generator().manual_seed(hidden)
    rows = 777
    x = torch.randn(rows, 2, generator=g)
    W1, b1 = torch.randn(16, 2, generator=g), torch.randn(16, generator=g)
    W2, b2 = torch.randn(hidden, 16, generator=g), torch.randn(hidden, generator=g)
    perm = torch.randperm(rows, generator=g).int()
    for gather in (None, perm):
        want = cpu_ops.encode(x.double(), W1.double(), b1.double(), W2.double(), b2.double(), gather=gather)
        got = ops.encode(x.to(dev()), W1.to(dev()), b1.to(dev()), W2.to(dev()), b2.to(dev()),
                         gather=None if gather is None else gather.to(dev()))
        _assert_close(got, want)


def test_encode_wide_hidden_ne_and_features():
    g = torch.Generator().manual_seed(5)
    x = torch.randn(300, 5, generator=g)
    W1, b1 = torch.randn(40, 5, generator=g), torch.randn(40, generator=g)
    W2, b2 = torch.randn(64, 40, generator=g), torch.randn(64, generator=g)
    want = cpu_ops.encode(x.double(), W1.double(), b1.double(), W2.double(), b2.double())
    _assert_close(ops.encode(*(t.to(dev()) for t in (x, W1, b1, W2, b2))), want)


@pytest.mark.parametrize("m,k,nout", [(1000, 64, 320), (333, 128, 640), (129, 256, 1280), (128, 64, 64), (5, 128, 32), (2000, 128, 96),
                                      (40_000, 128, 640), (33_333, 64, 320), (70_001, 128, 128), (1, 128, 256), (3, 128, 640), (31, 128, 1280),
                                      (20_001, 256, 1280), (9000, 256, 128)])   # K = 256 from 8192 rows: edge_gate_pl256.hip mode 4
def test_linear(m, k, nout):
    g = torch.Generator().manual_seed(m + k + nout)
    A, W, b = torch.randn(m, k, generator=g), torch.randn(nout, k, generator=g), torch.randn(nout, generator=g)
    want = A.double() @ W.double().t() + b.double()
    got = ops.linear(A.to(dev()), W.to(dev()), b.to(dev()))
    _assert_close(got, want, scale=float(k) ** 0.5 * 4)
    # asymmetric, transpose-detecting case: A = I-like selector
    A2 = torch.zeros(m, k)
    A2[torch.arange(m), torch.arange(m) % k] = 1.0
    got2 = ops.linear(A2.to(dev()), W.to(dev()), None)
    if (k == 128 and nout % 128 == 0 and nout >= 256) or (k == 256 and nout % 128 == 0):   # the fp16x3 route (node_project.hip): 1.0 * (w1 + w2 / 2048)
        assert torch.allclose(got2.cpu(), W.t()[torch.arange(m) % k], rtol=2.0 ** -21, atol=1e-9)   # is w to 2^-22, not bit for bit
    else:
        assert torch.equal(got2.cpu(), W.t()[torch.arange(m) % k])
    base = torch.randn(m, nout, generator=g)
    acc = ops.linear(A.to(dev()), W.to(dev()), b.to(dev()), out=base.to(dev()).clone(), accumulate=True)
    _assert_close(acc, want + base.double(), scale=float(k) ** 0.5 * 4)
    try:  # other kernels behind the same entry point: tile kernel (1), exact-fp32 weight-stationary (2), bf16x6 with LDS-staged A (3)
        ops.set_tuning(2, 5 if k == 128 else 4)   # the streaming kernel at any size
        stream = ops.linear(A.to(dev()), W.to(dev()), b.to(dev()))
        for variant in (1, 2, 3, 7, 8, 10, 9):   # 7 / 8: row-major 16-byte stores through an LDS tile / by in-register quad transposes
            ops.set_tuning(2, variant)       # 10: rounds 4-5's fp16x3 edge-tile kernels as the projection (edge_gate_bf.hip / edge_tile_f16.hip mode 4); 9: the default route by name
            _assert_close(ops.linear(A.to(dev()), W.to(dev()), b.to(dev())), want, scale=float(k) ** 0.5 * 4)
            if variant in (7, 8) and k in (64, 128) and nout % 64 == 0:   # the same arithmetic as the streaming kernel: the same bits
                assert torch.equal(ops.linear(A.to(dev()), W.to(dev()), b.to(dev())), stream)
            if variant == 9 and k == 128 and nout % 128 == 0 and nout >= 256:
                wide = torch.full((m, nout + 64), 7.0, device=dev())   # strided output, no bias; the default route at this shape
                ops.linear(A.to(dev()), W.to(dev()), None, out=wide[:, 64:])
                _assert_close(wide[:, 64:], want - b.double(), scale=float(k) ** 0.5 * 4)
                assert (wide[:, :64] == 7.0).all()
                assert torch.equal(ops.linear(A.to(dev()), W.to(dev()), b.to(dev())), got)
    finally:
        ops.set_tuning(2, 0)


@pytest.mark.parametrize("m,k,nout", [(400_001, 128, 640), (400_000, 64, 320), (1000, 128, 640), (129, 64, 256)])
def test_linear_a_stationary_kernel(m, k, nout):
    """k_linear_as (large M and a wide output: the node projection of a >= 400k-node graph; tuning variant 6 forces it at any
    size): the same bf16x6 products in the same order as the streaming kernel - bit-identical - and the strided-output,
    accumulate and ragged-last-block cases."""
    g = torch.Generator().manual_seed(m + k)
    A, W, b = torch.randn(m, k, generator=g).to(dev()), torch.randn(nout, k, generator=g).to(dev()), torch.randn(nout, generator=g).to(dev())
    try:
        ops.set_tuning(2, 6)
        got = ops.linear(A, W, b)
        wide = torch.zeros(m, nout + 64, device=dev())
        ops.linear(A, W, None, out=wide[:, 64:])
        base = torch.randn(m, nout, generator=g).to(dev())
        acc = ops.linear(A, W, b, out=base.clone(), accumulate=True)
        ops.set_tuning(2, 5 if k == 128 else 4)     # the streaming kernel at any size
        ref = ops.linear(A, W, b)
        ref_nb = ops.linear(A, W, None)
    finally:
        ops.set_tuning(2, 0)
    assert torch.equal(got, ref) and torch.equal(wide[:, 64:], ref_nb) and not wide[:, :64].any()
    if k == 64 or nout % 128:
        assert torch.equal(ops.linear(A, W, b), ref)  # the default route (A-stationary from 400k rows) agrees either way
    else:   # K = 128 and whole 128-column blocks: the default is the plane-form kernel at every size (its own order of the same products)
        assert (ops.linear(A, W, b) - ref).abs().max() <= 1e-4
    rows = torch.randint(0, m, (2000,), generator=g)
    want = A[rows].double().cpu() @ W.double().cpu().t() + b.double().cpu()
    _assert_close(got[rows], want, scale=float(k) ** 0.5 * 4)
    _assert_close(acc[rows], want + base[rows].double().cpu(), scale=float(k) ** 0.5 * 4)


def test_linear_strided_views():
    g = torch.Generator().manual_seed(9)
    hidden, hs, n = 128, 64, 700
    h = torch.randn(n, hidden, generator=g).to(dev())
    W1 = torch.randn(hs, 3 * hidden, generator=g).to(dev())
    b1 = torch.randn(hs, generator=g).to(dev())
    PQ = torch.zeros(n, 2 * hs, device=dev())
    ops.linear(h, W1[:, :hidden], None, out=PQ[:, :hs])
    ops.linear(h, W1[:, hidden:2 * hidden], b1, out=PQ[:, hs:])
    want = torch.cat([h.double() @ W1[:, :hidden].double().t(), h.double() @ W1[:, hidden:2 * hidden].double().t() + b1.double()], 1)
    _assert_close(PQ, want.cpu(), scale=50.0)


def _layer_inputs(hidden, n, e, seed):
    g = torch.Generator().manual_seed(seed)
    src, dst = _rand_graph(n, e, seed)
    t = {
        "e": 3.0 * torch.randn(e, hidden, generator=g), "h": torch.randn(n, hidden, generator=g),
        "P": torch.randn(n, 5 * hidden, generator=g), "W3": torch.randn(hidden, hidden, generator=g) / hidden ** 0.5,
        "scale": 0.5 + torch.rand(hidden, generator=g), "shift": torch.randn(hidden, generator=g),
    }
    return src, dst, t


@pytest.mark.parametrize("hidden", [64, 128, 256])
@pytest.mark.parametrize("norm", [0, 1])
@pytest.mark.parametrize("e_base", [1000, 70_001])  # one tile per workgroup / several tiles per persistent workgroup
def test_edge_gate(hidden, norm, e_base):
    n, e = 500, e_base + hidden  # not a multiple of the tile height
    src, dst, t = _layer_inputs(hidden, n, e, seed=hidden + norm)
    gv, cv = _views_pair(src, dst, n)
    H = hidden
    d = {k: v.to(dev()) for k, v in t.items()}
    e_sorted = t["e"]
    want = cpu_ops.edge_gate(e_sorted.double().clone(), t["P"][:, 3 * H:4 * H].double(), t["P"][:, 4 * H:].double(), cv,
                             t["W3"].double(), norm, t["scale"].double(), t["shift"].double())
    e_dev = d["e"].clone()
    ops.edge_gate(e_dev, d["P"][:, 3 * H:4 * H], d["P"][:, 4 * H:], gv, d["W3"], norm, d["scale"], d["shift"])  # in place
    _assert_close(e_dev, want, scale=20.0)
    out = torch.empty_like(d["e"])
    ops.edge_gate(d["e"], d["P"][:, 3 * H:4 * H], d["P"][:, 4 * H:], gv, d["W3"], norm, d["scale"], d["shift"], out=out)
    if hidden == 256 and norm == 0:   # separate buffers select the streaming kernel at H = 256, in place the tile kernel
        _assert_close(out, want, scale=20.0)
    else:
        assert torch.equal(out, e_dev)
    # every kernel variant behind the entry point (gnnome_set_tuning key 0) must meet the same contract
    try:
        for variant in (1, 5, 6, 7, 8):   # 8 = the second-generation bf16x6 kernel at H = 128 (the default there is its plane form)
            ops.set_tuning(0, variant)
            e_var = d["e"].clone()
            ops.edge_gate(e_var, d["P"][:, 3 * H:4 * H], d["P"][:, 4 * H:], gv, d["W3"], norm, d["scale"], d["shift"])
            _assert_close(e_var, want, scale=20.0)
    finally:
        ops.set_tuning(0, 0)


def test_h256_streaming_gate_in_pieces_equals_one_launch():
    """The H = 256 streaming gate cuts its launch into pieces of 64 tiles per workgroup (edge_gate_stream.hip): with the
    piece size forced down to 3 tiles (gnnome_set_tuning key 4) a ragged 300k-edge input runs as seven launches with
    offset row / index pointers - same bits as one uncut launch, which is itself checked against the contract."""
    H, n, e = 256, 3000, 300_007
    src, dst, t = _layer_inputs(H, n, e, seed=11)
    gv, cv = _views_pair(src, dst, n)
    d = {k: v.to(dev()) for k, v in t.items()}
    args = (d["e"], d["P"][:, 3 * H:4 * H], d["P"][:, 4 * H:], gv, d["W3"], 0, d["scale"], d["shift"])
    try:
        ops.set_tuning(0, 9)         # round 2's streaming kernel (the default at H = 256 is now the plane form, edge_gate_pl256.hip)
        ops.set_tuning(4, 1 << 20)   # one launch
        whole = ops.edge_gate(*args, out=torch.empty_like(d["e"]))
        ops.set_tuning(4, 3)
        pieces = ops.edge_gate(*args, out=torch.full_like(d["e"], float("nan")))
        ops.set_tuning(4, 0)
        stream_default = ops.edge_gate(*args, out=torch.empty_like(d["e"]))
    finally:
        ops.set_tuning(4, 0)
        ops.set_tuning(0, 0)
    assert torch.equal(pieces, whole) and torch.equal(stream_default, whole)
    # the wave-specialised plane form (W3 in registers, two workgroups per row) feeds the matrix cores the same k in the same order:
    # the same bits as the streaming kernel, at a ragged 300k edges
    try:
        ops.set_tuning(10, 1)    # (the default at H = 256 is round 4's fp16x3 kernel, edge_tile_f16.hip: tests/test_edge_tile_f16.py)
        default = ops.edge_gate(*args, out=torch.full_like(d["e"], float("nan")))
    finally:
        ops.set_tuning(10, 0)
    assert torch.equal(default, whole)
    # ... and as a residual GEMM C += A W^T (the backward's d e_in = d e' + dxe W3 at H = 256)
    A, C0 = d["e"], torch.randn_like(d["e"])
    c_new = ops.linear(A, d["W3"], None, out=C0.clone(), accumulate=True)
    try:
        ops.set_tuning(0, 9)
        c_old = ops.linear(A, d["W3"], None, out=C0.clone(), accumulate=True)
    finally:
        ops.set_tuning(0, 0)
    assert torch.equal(c_new, c_old)
    want = cpu_ops.edge_gate(t["e"].double().clone(), t["P"][:, 3 * H:4 * H].double(), t["P"][:, 4 * H:].double(), cv, t["W3"].double(), 0,
                             t["scale"].double(), t["shift"].double())
    _assert_close(whole, want, scale=20.0)


@pytest.mark.parametrize("hidden", [64, 128, 256])
@pytest.mark.parametrize("norm", [0, 1])
def test_node_aggregate(hidden, norm):
    n, e = 300, 2500
    src, dst, t = _layer_inputs(hidden, n, e, seed=7 * hidden + norm)
    # one hub node with a long in- and out-list
    src[:400], dst[400:800] = 5, 5
    gv, cv = _views_pair(src, dst, n)
    H = hidden
    d = {k: v.to(dev()) for k, v in t.items()}
    P64 = t["P"].double()
    want = cpu_ops.node_aggregate(t["e"].double(), P64[:, :H], P64[:, H:2 * H], P64[:, 2 * H:3 * H], cv, t["h"].double(), norm,
                                  t["scale"].double(), t["shift"].double())
    got = ops.node_aggregate(d["e"], d["P"][:, :H], d["P"][:, H:2 * H], d["P"][:, 2 * H:3 * H], gv, d["h"], norm, d["scale"], d["shift"])
    _assert_close(got, want, scale=10.0)
    # isolated nodes: fwd = bwd = 0 exactly
    iso = n - 1
    a1 = P64[iso, :H]
    if norm == 0:
        exp = torch.relu(a1 * t["scale"].double() + t["shift"].double()) + t["h"][iso].double()
        _assert_close(got[iso], exp)
    # partial update (destination-range partition): only the first rows are written
    got_part = ops.node_aggregate(d["e"], d["P"][:, :H], d["P"][:, H:2 * H], d["P"][:, 2 * H:3 * H], gv, d["h"], norm, d["scale"],
                                  d["shift"], num_nodes_out=100)
    assert torch.equal(got_part[:100], got[:100])


@pytest.mark.parametrize("hidden,deg", [(128, 100_000), (64, 5000), (256, 20_000)])
def test_node_aggregate_hub_split_path(hidden, deg):
    """SURVEY.md 7 "Skew": a node with `deg` in-edges and `deg` out-edges (a repeat-induced hub).  Its list is split into
    fixed chunks reduced by separate waves and added in chunk order: same result as the contract to fp32 accuracy, the same
    bits on every run, and the other nodes untouched by the change of path."""
    import time
    n = 3000
    g = torch.Generator().manual_seed(deg + hidden)
    e_bg = 20_000
    src = torch.randint(0, n - 3, (e_bg + 2 * deg,), generator=g).int()
    dst = torch.randint(0, n - 3, (e_bg + 2 * deg,), generator=g).int()
    hub = 17
    dst[e_bg:e_bg + deg] = hub          # deg in-edges
    src[e_bg + deg:] = hub              # deg out-edges
    E, H = src.numel(), hidden
    t = {"e": 2.0 * torch.randn(E, H, generator=g), "h": torch.randn(n, H, generator=g), "P": torch.randn(n, 3 * H, generator=g),
         "scale": 0.5 + torch.rand(H, generator=g), "shift": torch.randn(H, generator=g)}
    gv, cv = _views_pair(src, dst, n)
    d = {k: v.to(dev()) for k, v in t.items()}
    P64 = t["P"].double()
    want = cpu_ops.node_aggregate(t["e"].double(), P64[:, :H], P64[:, H:2 * H], P64[:, 2 * H:], cv, t["h"].double(), 0, t["scale"].double(),
                                  t["shift"].double())
    run = lambda: ops.node_aggregate(d["e"], d["P"][:, :H], d["P"][:, H:2 * H], d["P"][:, 2 * H:], gv, d["h"], 0, d["scale"], d["shift"])  # noqa: E731
    got = run()
    _assert_close(got, want, scale=10.0)
    assert torch.equal(got, run())                                           # bit-reproducible
    try:   # the single-wave path (hub split off): every OTHER node has the same bits, the hub the same value to rounding
        ops.set_tuning(6, 1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        single = run()
        torch.cuda.synchronize()
        t_single = time.perf_counter() - t0
    finally:
        ops.set_tuning(6, 0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run()
    torch.cuda.synchronize()
    t_split = time.perf_counter() - t0
    mask = torch.ones(n, dtype=torch.bool)
    mask[hub] = False
    assert torch.equal(got[mask.to(dev())], single[mask.to(dev())])
    _assert_close(single[hub], want[hub], scale=10.0)
    print(f"hub of 2 x {deg} edges, H={hidden}: split path {t_split * 1e3:.3f} ms, single wave {t_single * 1e3:.3f} ms")
    # two streams at once (round 4: the hub scratch is per (device, stream), it used to be shared behind a "don't" contract): a second
    # graph with its own hub aggregates on a side stream while this one runs on the current stream - both equal their solo results
    src2, dst2 = src.clone(), dst.clone()
    dst2[e_bg:e_bg + deg], src2[e_bg + deg:] = 99, 99
    gv2 = ops.GraphViews(src2.to(dev()), dst2.to(dev()), n)
    run2 = lambda: ops.node_aggregate(d["e"], d["P"][:, :H], d["P"][:, H:2 * H], d["P"][:, 2 * H:], gv2, d["h"], 0, d["scale"], d["shift"])  # noqa: E731
    solo2 = run2()
    side = torch.cuda.Stream(device=dev())
    torch.cuda.synchronize()
    outs = []
    for _ in range(4):
        with torch.cuda.stream(side):
            b = run2()
        a_ = run()
        outs.append((a_, b))
    torch.cuda.synchronize()
    assert all(torch.equal(a_, got) and torch.equal(b, solo2) for a_, b in outs)


@pytest.mark.parametrize("hidden", [64, 128, 256])
def test_node_aggregate_in_node_ranges_is_bit_identical(hidden):
    """gnnome_node_aggregate_range_f32: one aggregation cut into consecutive node ranges (ragged, one of them a single row,
    a 6000-edge hub in the SECOND range - the hub path runs with the range that starts at node 0) gives the bits of the
    single launch, and rows outside a range are not touched."""
    n, H = 5003, hidden
    g = torch.Generator().manual_seed(hidden)
    src, dst = _rand_graph(n, 60_000, seed=hidden)
    hub = 3100
    dst[:6000] = hub
    src[6000:9000] = hub
    E = src.numel()
    d = {"e": (2.0 * torch.randn(E, H, generator=g)).to(dev()), "h": torch.randn(n, H, generator=g).to(dev()),
         "P": torch.randn(n, 3 * H, generator=g).to(dev()), "scale": (0.5 + torch.rand(H, generator=g)).to(dev()),
         "shift": torch.randn(H, generator=g).to(dev())}
    gv, _ = _views_pair(src, dst, n)
    args = (d["e"], d["P"][:, :H], d["P"][:, H:2 * H], d["P"][:, 2 * H:], gv, d["h"], 0, d["scale"], d["shift"])
    whole = ops.node_aggregate(*args)
    out = torch.full_like(whole, float("nan"))
    bounds = [0, 1, 1500, 1501, 4096, n]
    for lo, hi in zip(bounds[:-1], bounds[1:]):
        ops.node_aggregate(*args, node_range=(lo, hi), out=out)
        untouched = torch.arange(n, device=dev()) >= hi
        untouched[hub] = False   # the hub's row is written by the first range's hub launch
        assert torch.isnan(out[untouched]).all() and not torch.isnan(out[:hi]).any()
    assert torch.equal(out, whole)
    with pytest.raises(Exception):
        ops.node_aggregate(*args, node_range=(10, 5), out=out)


def test_pipelined_projection_is_bit_identical_and_capturable():
    """engine.aggregate_then_project (the next layer's node projection on a second stream under the aggregation's node
    ranges; an option, off by default because it measured slower): same logits bit for bit as the one-stream sequence,
    eager and replayed from a hipGraph, run after run."""
    from gnnome_amd import engine
    from gnnome_amd.capture import CapturedForward
    n, e, hidden = 40_000, 400_000, 128
    gr = make_graph(n, e, seed=5, kind="banded")
    x = degree_features(gr["src"], gr["dst"], n).to(dev())
    ef = gr["e"].to(dev())
    m = _model(random_state_dict(hidden, seed=5), hidden)
    views = views_for((gr["src"].to(dev()), gr["dst"].to(dev()), n), dev())
    keep = engine.PIPELINE_CHUNKS, engine.PIPELINE_MIN_NODES
    try:
        engine.PIPELINE_CHUNKS = 1
        plain = m(views, x, ef)
        for chunks in (2, 4, 7):
            engine.PIPELINE_CHUNKS, engine.PIPELINE_MIN_NODES = chunks, 0
            for _ in range(3):
                assert torch.equal(m(views, x, ef), plain), chunks
        engine.PIPELINE_CHUNKS = 4
        cap = CapturedForward(m, views, x, ef)
        for _ in range(3):
            assert torch.equal(cap(), plain)
    finally:
        engine.PIPELINE_CHUNKS, engine.PIPELINE_MIN_NODES = keep


@pytest.mark.parametrize("hidden,hs", [(64, 64), (128, 64), (256, 64), (64, 32), (128, 128)])
def test_edge_score(hidden, hs):
    n, e = 400, 1500
    g = torch.Generator().manual_seed(hidden + hs)
    src, dst = _rand_graph(n, e, hidden + hs)
    gv, cv = _views_pair(src, dst, n)
    t = {
        "e": 2.0 * torch.randn(e, hidden, generator=g), "PQ": torch.randn(n, 2 * hs, generator=g),
        "W1": torch.randn(hs, 3 * hidden, generator=g) / hidden ** 0.5, "W2": torch.randn(32, hs, generator=g) / hs ** 0.5,
        "b2": torch.randn(32, generator=g), "W3": torch.randn(32, generator=g), "b3": torch.randn(1, generator=g),
    }
    d = {k: v.to(dev()) for k, v in t.items()}
    t64 = {k: v.double() for k, v in t.items()}
    for scatter in (True, False):
        want = cpu_ops.edge_score(t64["e"], t64["PQ"][:, :hs], t64["PQ"][:, hs:], cv, t64["W1"][:, 2 * hidden:], t64["W2"], t64["b2"],
                                  t64["W3"], t64["b3"], torch.zeros(e, dtype=torch.float64), scatter_to_edge_id=scatter)
        got = ops.edge_score(d["e"], d["PQ"][:, :hs], d["PQ"][:, hs:], gv, d["W1"][:, 2 * hidden:], d["W2"], d["b2"], d["W3"], d["b3"],
                             torch.zeros(e, device=dev()), scatter_to_edge_id=scatter)
        _assert_close(got, want, scale=20.0)


@pytest.mark.parametrize("hidden,e", [(128, 1), (128, 33), (128, 400_003), (64, 70_001), (128, 8 * 32 * 256 + 5)])
def test_edge_score_streaming_kernel_against_the_tile_kernel(hidden, e):
    """k_edge_score_ws (inference, hs = 64: W1e stationary in LDS, e rows streamed into MFMA fragments, wave-private tail)
    against the 128-edge tile kernel behind the same entry point (gnnome_set_tuning(2, 1)): one edge, a ragged last tile,
    fewer tiles than waves, more tiles than one round of the persistent grid; scattered and sorted output; repeatable."""
    hs, n = 64, max(8, e // 9)
    g = torch.Generator().manual_seed(hidden + e)
    src, dst = _rand_graph(n, e, hidden + e)
    gv, _ = _views_pair(src, dst, n)
    ee = (2.0 * torch.randn(e, hidden, generator=g)).to(dev())
    PQ = torch.randn(n, 2 * hs, generator=g).to(dev())
    W1e = (torch.randn(hs, hidden, generator=g) / hidden ** 0.5).to(dev())
    W2, b2 = (torch.randn(32, hs, generator=g) / hs ** 0.5).to(dev()), torch.randn(32, generator=g).to(dev())
    W3, b3 = torch.randn(32, generator=g).to(dev()), torch.randn(1, generator=g).to(dev())
    for scatter in (True, False):
        run = lambda: ops.edge_score(ee, PQ[:, :hs], PQ[:, hs:], gv, W1e, W2, b2, W3, b3, torch.zeros(e, device=dev()), scatter_to_edge_id=scatter)  # noqa: E731
        got = run()
        try:
            ops.set_tuning(2, 1)
            ref = run()
        finally:
            ops.set_tuning(2, 0)
        assert (got - ref).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item())
        assert torch.equal(got, run())


@pytest.mark.parametrize("hidden", [64, 128, 256])   # 256: round 4, mode 5 of the fp16x3 edge-tile kernel
@pytest.mark.parametrize("e_count", [5, 777, 90_001])
def test_edge_gate_with_folded_encoder(hidden, e_count):
    n, H = 600, hidden
    src, dst, t = _layer_inputs(hidden, n, e_count, seed=hidden + 5)
    gv, cv = _views_pair(src, dst, n)
    g = torch.Generator().manual_seed(3)
    e_raw = torch.stack([torch.randn(e_count, generator=g), 0.9 + 0.1 * torch.rand(e_count, generator=g)], 1)
    enc = (torch.randn(16, 2, generator=g), torch.randn(16, generator=g), torch.randn(H, 16, generator=g) / 4, torch.randn(H, generator=g))
    e0 = cpu_ops.encode(e_raw.double(), *(w.double() for w in enc), gather=cv.srt_eid)
    want = cpu_ops.edge_gate(e0.clone(), t["P"][:, 3 * H:4 * H].double(), t["P"][:, 4 * H:].double(), cv, t["W3"].double(), 0,
                             t["scale"].double(), t["shift"].double())
    d = {k: v.to(dev()) for k, v in t.items()}
    enc_d = tuple(w.to(dev()) for w in enc)
    assert ops.can_fuse_edge_encoder(e_raw.to(dev()), enc_d, H, 0, d["P"][:, 3 * H:4 * H])
    got = ops.edge_gate_encode(e_raw.to(dev()), enc_d, d["P"][:, 3 * H:4 * H], d["P"][:, 4 * H:], gv, d["W3"], d["scale"], d["shift"])
    _assert_close(got, want, scale=20.0)
    assert torch.equal(got, ops.edge_gate_encode(e_raw.to(dev()), enc_d, d["P"][:, 3 * H:4 * H], d["P"][:, 4 * H:], gv, d["W3"], d["scale"], d["shift"]))
    if hidden == 256:   # against the unfused pair on the same device: encoder, then the gate
        e0d = ops.encode(e_raw.to(dev()), *enc_d, gather=gv.srt_eid)
        two = ops.edge_gate(e0d, d["P"][:, 3 * H:4 * H], d["P"][:, 4 * H:], gv, d["W3"], 0, d["scale"], d["shift"], out=torch.empty_like(e0d))
        assert (got - two).abs().max().item() <= 2e-5 * max(1.0, two.abs().max().item())
        return
    alts = []
    try:  # the exact-fp32-MFMA generation of the kernel, slot hand-over by LDS counters (5) or workgroup barriers (6);
          # 8 = the second-generation bf16x6 kernel with the folded encoder (the default is its plane form)
        for variant in (5, 6, 8):
            ops.set_tuning(0, variant)
            alts.append(ops.edge_gate_encode(e_raw.to(dev()), enc_d, d["P"][:, 3 * H:4 * H], d["P"][:, 4 * H:], gv, d["W3"], d["scale"],
                                             d["shift"]))
    finally:
        ops.set_tuning(0, 0)
    assert torch.equal(alts[0], alts[1])          # same arithmetic, same bits
    _assert_close(alts[0], want, scale=20.0)
    _assert_close(alts[2], want, scale=20.0)


def test_gather_rows():
    g = torch.Generator().manual_seed(2)
    table = torch.randn(100, 128, generator=g).to(dev())
    idx = torch.randint(0, 100, (257,), generator=g).int().to(dev())
    assert torch.equal(ops.gather_rows(table, idx), table[idx.long()])


# ------------------------------------------------------------------------------------ whole model

def _model(sd, hidden, normalization="batch", layers=8):
    m = gnnome_amd.models.SymGatedGCNModel(2, 2, hidden, 16, layers, sd["predictor.W1.weight"].shape[0], normalization).eval()
    m.load_state_dict(sd)
    return m.to(dev())


def _prob_diff(a, b):
    return (torch.sigmoid(a.double().cpu()) - torch.sigmoid(b.double().cpu())).abs().max().item()


def test_goldens_shipped_weights(shipped_weights):
    m = _model(shipped_weights, 64)
    for name in ("g1_hand.pt", "g2_uniform_1k.pt"):
        g = load_golden(name)
        out = m((g["src"], g["dst"], g["num_nodes"]), g["x"].to(dev()), g["e"].to(dev()))
        assert out.shape == g["logits"].shape and out.dtype == torch.float32 and out.is_cuda
        assert _prob_diff(out, g["logits"]) < PROB_TOL
    # CPU inputs (inference.py:388 pins device='cpu'): staged in, logits returned on the CPU
    g = load_golden("g2_uniform_1k.pt")
    x0, e0 = g["x"].clone(), g["e"].clone()
    out = m((g["src"], g["dst"], g["num_nodes"]), g["x"], g["e"])
    assert out.device.type == "cpu" and _prob_diff(out, g["logits"]) < PROB_TOL
    assert torch.equal(g["x"], x0) and torch.equal(g["e"], e0)  # inputs untouched


def test_against_the_fp64_c_oracle(shipped_weights):
    """The HIP path against the double-precision C restatement: the 1e-4 bar measured from the truth, not from
    another fp32 evaluation (the reference's own fp32 output is 4.9e-5 from it on this graph)."""
    from oracle import c_oracle
    g = load_golden("g2_uniform_1k.pt")
    truth = c_oracle.forward(shipped_weights, g["src"], g["dst"], g["num_nodes"], g["x"], g["e"], precision="f64")
    out = _model(shipped_weights, 64)((g["src"], g["dst"], g["num_nodes"]), g["x"].to(dev()), g["e"].to(dev()))
    assert _prob_diff(out, truth) < PROB_TOL


def test_goldens_wider_hidden_and_layernorm():
    for hidden in (128, 256):
        g = load_golden(f"g5_eval_h{hidden}.pt")
        m = _model(random_state_dict(hidden, seed=g["seed"]), hidden)
        out = m((g["src"], g["dst"], g["num_nodes"]), g["x"].to(dev()), g["e"].to(dev()))
        assert _prob_diff(out, g["logits"]) < PROB_TOL
    g = load_golden("g6_layernorm_h64.pt")
    sd = {k: v for k, v in random_state_dict(64, seed=g["seed"]).items() if "running_" not in k and "num_batches" not in k}
    out = _model(sd, 64, "layer")((g["src"], g["dst"], g["num_nodes"]), g["x"].to(dev()), g["e"].to(dev()))
    assert _prob_diff(out, g["logits"]) < PROB_TOL


def test_layernorm_at_widths_between_the_built_ones_golden_g12():
    """normalization='layer' at (96, 48) and (160, 128): the gate / aggregation kernels' LayerNorm over the model's own channels
    (GNNOME_NORM_LAYER_OVER) against the reference's logits."""
    g = load_golden("g12_layernorm_widths.pt")
    for case in g["cases"]:
        sd = {k: v for k, v in random_state_dict(case["hidden"], num_layers=case["layers"], hidden_edge_scores=case["hs"], seed=case["seed"]).items()
              if "running_" not in k and "num_batches" not in k}
        m = gnnome_amd.SymGatedGCNModel(2, 2, case["hidden"], 16, case["layers"], case["hs"], "layer").eval()
        m.load_state_dict(sd)
        m.to(dev())
        out = m((g["src"], g["dst"], g["num_nodes"]), g["x"].to(dev()), g["e"].to(dev()))
        assert _prob_diff(out, case["eval_logits"]) < 2e-6 * 50    # (reference-exact up to fp32 reordering)


def test_widths_between_the_built_ones():
    """hidden_features / hidden_edge_scores the kernels are not built for (the reference takes any, configs/hyperparameters.py:22-24) run on the
    next built width with zero-padded parameters (engine.BUILT_HIDDEN): the reference's own logits (golden G11), the oracle at 20k / 200k, the
    layer- and predictor-level entries at their reference shapes, the hipGraph replay and the partitioned runner.  (Train mode: tests/test_hip_training.py.)"""
    from gnnome_amd import dist as gdist
    from gnnome_amd.capture import CapturedForward
    g = load_golden("g11_widths.pt")
    graph = (g["src"], g["dst"], g["num_nodes"])
    for case in g["cases"]:
        sd = random_state_dict(case["hidden"], num_layers=case["layers"], hidden_edge_scores=case["hs"], seed=case["seed"])
        m = _model(sd, case["hidden"], layers=case["layers"])
        out = m(graph, g["x"].to(dev()), g["e"].to(dev()))
        assert out.shape == case["logits"].shape and _prob_diff(out, case["logits"]) < PROB_TOL
        cap = CapturedForward(m, graph, g["x"].to(dev()), g["e"].to(dev()))
        assert torch.equal(cap(), out)
    # layer- and predictor-level API: [N,96] / [E,96] rows in and out, against the oracle's layer
    hidden, hs = 96, 48
    sd = random_state_dict(hidden, num_layers=2, hidden_edge_scores=hs, seed=3)
    om = model_from_state_dict(sd).eval()
    m = _model(sd, hidden, layers=2)
    gen = torch.Generator().manual_seed(2)
    h = torch.randn(g["num_nodes"], hidden, generator=gen)
    e = torch.randn(g["src"].numel(), hidden, generator=gen)
    with torch.no_grad():
        wh, we = om._layer_forward(om.gnn.convs[0], g["src"].long(), g["dst"].long(), g["num_nodes"], h, e)
        ws = om._score(g["src"].long(), g["dst"].long(), h, e)
        gh, ge = m.gnn.convs[0](graph, h.to(dev()), e.to(dev()))
        gs = m.predictor(graph, h.to(dev()), e.to(dev()))
    assert gh.shape == wh.shape and ge.shape == we.shape and gs.shape == ws.shape
    _assert_close(gh, wh.double(), tol=2e-5)
    _assert_close(ge, we.double(), tol=2e-5)
    _assert_close(gs, ws.double(), tol=2e-5)
    # mid size against the oracle, and the same model through the partitioned kernel sequence (dist.py)
    n, ec = 20000, 200000
    gr = make_graph(n, ec, seed=4, kind="banded")
    x = degree_features(gr["src"], gr["dst"], n)
    for hidden, hs in ((96, 48), (200, 100)):
        sd = random_state_dict(hidden, num_layers=4, hidden_edge_scores=hs, seed=5)
        with torch.no_grad():
            want = model_from_state_dict(sd).eval()((gr["src"], gr["dst"], n), x, gr["e"])
        m = _model(sd, hidden, layers=4)
        got = m((gr["src"], gr["dst"], n), x.to(dev()), gr["e"].to(dev()))
        assert _prob_diff(got, want) < PROB_TOL
        part = gdist.PartitionedGraph.from_global(gr["src"], gr["dst"], n, 0, 1, dev())
        assert torch.equal(gdist.PartitionedRunner(m, part, x, gr["e"], dev()).forward(), got)


def test_golden_reversed_graph_both_ways():
    g = load_golden("g4_reverse_h64.pt")
    m = _model(random_state_dict(64, seed=g["seed"]), 64)
    x, xr, e = g["x"].to(dev()), g["x_rev"].to(dev()), g["e"].to(dev())
    org = m((g["src"], g["dst"], g["num_nodes"]), x, e)
    assert _prob_diff(org, g["logits"]) < PROB_TOL
    rev_rebuilt = m((g["dst"], g["src"], g["num_nodes"]), xr, e)              # what dgl.reverse hands over
    rev_free = m(reverse((g["src"], g["dst"], g["num_nodes"]), dev()), xr, e)  # same views, transposed
    assert _prob_diff(rev_rebuilt, g["logits_rev"]) < PROB_TOL
    assert _prob_diff(rev_free, g["logits_rev"]) < PROB_TOL


def test_hipgraph_capture_replays_bit_identically(shipped_weights):
    from gnnome_amd.capture import CapturedForward
    g = load_golden("g2_uniform_1k.pt")
    m = _model(shipped_weights, 64)
    x, e = g["x"].to(dev()), g["e"].to(dev())
    eager = m((g["src"], g["dst"], g["num_nodes"]), x, e)
    cap = CapturedForward(m, (g["src"], g["dst"], g["num_nodes"]), x, e)
    assert torch.equal(cap(), eager) and torch.equal(cap(), eager)
    e2 = e * 0.5
    assert torch.equal(cap(e=e2).clone(), m((g["src"], g["dst"], g["num_nodes"]), x, e2))


class _DuckGraph:
    """Anything with edges()/num_nodes()/num_edges() is accepted in place of a DGLGraph."""

    def __init__(self, src, dst, n):
        self._s, self._d, self._n = src, dst, n
        self.ndata, self.edata = {"keep": 1}, {"keep": 2}

    def edges(self):
        return self._s.long(), self._d.long()

    def num_nodes(self):
        return self._n

    def num_edges(self):
        return self._s.numel()


def test_duck_typed_graph_cache_and_no_mutation(shipped_weights):
    g = load_golden("g2_uniform_1k.pt")
    graph = _DuckGraph(g["src"], g["dst"], g["num_nodes"])
    m = _model(shipped_weights, 64)
    a = m(graph, g["x"].to(dev()), g["e"].to(dev()))
    v1 = views_for(graph, dev())
    b = m(graph, g["x"].to(dev()), g["e"].to(dev()))
    assert views_for(graph, dev()) is v1            # cached per graph object
    assert torch.equal(a, b)                        # deterministic, bit for bit
    assert graph.ndata == {"keep": 1} and graph.edata == {"keep": 2}
    assert _prob_diff(a, g["logits"]) < PROB_TOL


def test_layer_and_predictor_level_api(shipped_weights):
    g = load_golden("g2_uniform_1k.pt")
    om = model_from_state_dict(shipped_weights).eval()
    m = _model(shipped_weights, 64)
    gen = torch.Generator().manual_seed(1)
    h = torch.randn(g["num_nodes"], 64, generator=gen)
    e = torch.randn(g["src"].numel(), 64, generator=gen)
    graph = (g["src"], g["dst"], g["num_nodes"])
    with torch.no_grad():
        wh, we = om._layer_forward(om.gnn.convs[0], g["src"].long(), g["dst"].long(), g["num_nodes"], h, e)
        ws = om._score(g["src"].long(), g["dst"].long(), h, e)
    # the layer-level entries are inference-only: with grad enabled and trainable parameters they REFUSE (their outputs would
    # carry no autograd history and a fine-tune through them would train nothing) - ADVICE r2
    with pytest.raises(NotImplementedError):
        m.gnn.convs[0](graph, h.to(dev()), e.to(dev()))
    with pytest.raises(NotImplementedError):
        m.predictor(graph, h.to(dev()), e.to(dev()))
    with torch.no_grad():
        gh, ge = m.gnn.convs[0](graph, h.to(dev()), e.to(dev()))
        gs = m.predictor(graph, h.to(dev()), e.to(dev()))
        ph, pe = m.gnn(graph, h.to(dev()), e.to(dev()))  # processor loop, layers/processor.py:16-19
    _assert_close(gh, wh.double(), tol=2e-5)
    _assert_close(ge, we.double(), tol=2e-5)
    _assert_close(gs, ws.double(), tol=2e-5)
    assert ph.shape == h.shape and pe.shape == e.shape and torch.isfinite(ph).all()


def test_empty_graph(shipped_weights):
    m = _model(shipped_weights, 64)
    out = m((torch.zeros(0, dtype=torch.int32), torch.zeros(0, dtype=torch.int32), 6), torch.randn(6, 2).to(dev()), torch.zeros(0, 2).to(dev()))
    assert out.shape == (0, 1)


def test_partitioned_runner_world1_equals_plain_forward():
    """The dist.py kernel sequence (owned/halo split, prefix scoring, global-id scatter) on the HIP backend."""
    from gnnome_amd import dist as gdist
    n, e, hidden = 30000, 300000, 128
    gr = make_graph(n, e, seed=7, kind="banded")
    x = degree_features(gr["src"], gr["dst"], n)
    m = _model(random_state_dict(hidden, seed=1), hidden)
    part = gdist.PartitionedGraph.from_global(gr["src"], gr["dst"], n, 0, 1, dev())
    assert part.n_own == n and part.n_score == e and part.n_local == n
    got = gdist.PartitionedRunner(m, part, x, gr["e"], dev()).forward()
    want = m((gr["src"], gr["dst"], n), x.to(dev()), gr["e"].to(dev()))
    assert torch.equal(got, want)


# ------------------------------------------------------------------------------------ vs oracle, mid size

@pytest.mark.parametrize("hidden,kind", [(64, "banded"), (128, "banded"), (128, "uniform"), (256, "banded")])
def test_oracle_mid_size(hidden, kind):
    n, e = 20000, 200000
    gr = make_graph(n, e, seed=1, kind=kind)
    x = degree_features(gr["src"], gr["dst"], n)
    sd = random_state_dict(hidden, seed=1)
    om = model_from_state_dict(sd).eval()
    with torch.no_grad():
        want = om((gr["src"], gr["dst"], n), x, gr["e"])
    got = _model(sd, hidden)((gr["src"], gr["dst"], n), x.to(dev()), gr["e"].to(dev()))
    assert _prob_diff(got, want) < PROB_TOL


# ------------------------------------------------------------------------------------ full size properties

def _swap_roles(sd, hidden):
    """Weights of the model that sees dgl.reverse(g): A_2<->A_3, B_1<->B_2, predictor src/dst blocks swapped."""
    out = dict(sd)
    for k in sd:
        for a, b in (("A_2", "A_3"), ("B_1", "B_2")):
            if f".{a}." in k:
                out[k], out[k.replace(a, b)] = sd[k.replace(a, b)], sd[k]
    w = sd["predictor.W1.weight"]
    out["predictor.W1.weight"] = torch.cat([w[:, hidden:2 * hidden], w[:, :hidden], w[:, 2 * hidden:]], 1)
    return out


@pytest.mark.parametrize("n,e,hidden", [(100_000, 1_000_000, 128), (1_000_000, 10_000_000, 128)])
def test_full_size_properties(n, e, hidden):
    """BASELINE configs[1] (1M edges, H=128) and the 10M-edge target graph: properties that need no oracle."""
    gr = make_graph(n, e, seed=1, kind="banded")
    x = degree_features(gr["src"], gr["dst"], n).to(dev())
    ef = gr["e"].to(dev())
    sd = random_state_dict(hidden, seed=1)
    m = _model(sd, hidden)
    graph = (gr["src"].to(dev()), gr["dst"].to(dev()), n)
    views = views_for(graph, dev())
    a = m(views, x, ef)
    assert a.shape == (e, 1) and torch.isfinite(a).all()
    assert torch.equal(a, m(views, x, ef))                                       # run-to-run bit identical
    # mates (u,v) / (v^1,u^1) carry equal features; on this strand-symmetric graph x differs, so only
    # check the transposition identity: model_swapped(reverse(g), x, e) == model(g, x, e)
    ms = _model(_swap_roles(sd, hidden), hidden)
    b = ms(views.reversed(), x, ef)
    assert _prob_diff(a, b) < PROB_TOL
    c = ms((graph[1], graph[0], n), x, ef)                                       # rebuilt views of the reversed list
    assert _prob_diff(a, c) < PROB_TOL
    # edge-id permutation equivariance
    perm = torch.randperm(e, generator=torch.Generator().manual_seed(3)).to(dev())
    d = m((graph[0][perm], graph[1][perm], n), x, ef[perm])
    assert _prob_diff(a[perm], d) < PROB_TOL


@pytest.mark.parametrize("kind", ["banded", "uniform"])
def test_gate_plane_form_at_full_size_against_the_second_generation(kind):
    """k_edge_gate_pl hands an LDS slot over four times per tile (planes -> all compute waves read -> x tile -> both store waves
    drained); a missing hand-over shows up as a handful of wrong rows per MILLION edges (seen while it was written: 5 rows, rows
    28-31 of their tiles), which no small test catches.  1M edges, H = 128, five launches: every launch equals the
    second-generation kernel (variant 8) to fp32 reordering, in place and out of place, and repeats bit for bit."""
    n, e, H = 100_000, 1_000_000, 128
    gr = make_graph(n, e, seed=4, kind=kind)
    gv = ops.GraphViews(gr["src"].to(dev()), gr["dst"].to(dev()), n)
    g = torch.Generator(device=dev()).manual_seed(1)
    ee = torch.randn(e, H, device=dev(), generator=g)
    P = torch.randn(n, 2 * H, device=dev(), generator=g)
    W3 = torch.randn(H, H, device=dev(), generator=g) / H ** 0.5
    sc, sh = torch.rand(H, device=dev(), generator=g) * 0.1, torch.randn(H, device=dev(), generator=g)
    try:
        ops.set_tuning(0, 8)
        ref = ops.edge_gate(ee, P[:, :H], P[:, H:], gv, W3, 0, sc, sh, out=torch.empty_like(ee))
    finally:
        ops.set_tuning(0, 0)
    scale = ref.abs().max().item()
    first = None
    for rep in range(5):
        out = ops.edge_gate(ee, P[:, :H], P[:, H:], gv, W3, 0, sc, sh, out=torch.empty_like(ee))
        assert (out - ref).abs().max().item() <= 2e-6 * scale
        first = out if first is None else first
        assert torch.equal(out, first)
    inplace = ee.clone()
    ops.edge_gate(inplace, P[:, :H], P[:, H:], gv, W3, 0, sc, sh)
    assert torch.equal(inplace, first)
    # the training modes of the plane form: raw gate + statistics (mode 1) and BatchNorm backward + data gradient (mode 3)
    a_, c1, c2 = torch.rand(H, device=dev(), generator=g) + 0.5, 0.1 * sh, 0.1 * sc
    mean, rstd = sh * 0.5, torch.rand(H, device=dev(), generator=g) + 0.5

    def raw_gate():
        xe, (d1, d2, c, rows) = ops.edge_gate_raw_moments(ee, P[:, :H], P[:, H:], gv, W3)
        return xe, d1, d2

    def bn_backward(xe):   # (the relu mask is discontinuous in xe: both kernels get the SAME xe)
        C = ee.clone()
        dxe = ops.bn_bwd_dgrad(C, xe, sc + 0.5, sh, a_, c1, c2, mean, rstd, W3)
        return dxe, C
    try:
        ops.set_tuning(0, 8)
        want_raw = raw_gate()
        want_bwd = bn_backward(want_raw[0])
    finally:
        ops.set_tuning(0, 0)
    for rep in range(3):
        got_raw, got_bwd = raw_gate(), bn_backward(want_raw[0])
        for gt, wt, tol in zip(got_raw + got_bwd, want_raw + want_bwd, (2e-6, 2e-5, 2e-5, 2e-6, 2e-6)):
            assert (gt - wt).abs().max().item() <= tol * max(1.0, wt.abs().max().item())
        assert all(torch.equal(u, v) for u, v in zip(got_raw + got_bwd, raw_gate() + bn_backward(want_raw[0])))
    # layer 0's form (k_edge_gate_enc16: the encoder folded algebraically, K = 16) against the second generation's, which
    # computes e0 in fp32 and runs the K = 128 product on it: equal up to the reassociation (W3 W2) t vs W3 (W2 t)
    e_raw = gr["e"].to(dev())
    enc = (torch.randn(16, 2, device=dev(), generator=g), torch.randn(16, device=dev(), generator=g),
           torch.randn(H, 16, device=dev(), generator=g) / 4, torch.randn(H, device=dev(), generator=g))
    try:
        ops.set_tuning(0, 8)
        ref = ops.edge_gate_encode(e_raw, enc, P[:, :H], P[:, H:], gv, W3, sc, sh)
    finally:
        ops.set_tuning(0, 0)
    scale = ref.abs().max().item()
    first = None
    for rep in range(5):
        out = ops.edge_gate_encode(e_raw, enc, P[:, :H], P[:, H:], gv, W3, sc, sh)
        assert (out - ref).abs().max().item() <= 4e-6 * scale
        first = out if first is None else first
        assert torch.equal(out, first)


@pytest.mark.parametrize("n,e", [(250_000, 2_500_000), (2_000_000, 20_000_000)])
def test_h256_configs3_properties(n, e):
    """BASELINE configs[3] at H = 256 - one GPU's eighth of it (N = 250k, E = 2.5M), and the WHOLE graph (N = 2M, E = 20M:
    20.5 GB of edge state, twice for the streaming gate's two buffers - one MI355X holds it): run-to-run identical, the
    reversed-graph / swapped-roles identity, edge-id permutation equivariance - the size-independent properties of the path."""
    hidden = 256
    gr = make_graph(n, e, seed=1, kind="banded")
    x = degree_features(gr["src"], gr["dst"], n).to(dev())
    ef = gr["e"].to(dev())
    sd = random_state_dict(hidden, seed=1)
    m = _model(sd, hidden)
    graph = (gr["src"].to(dev()), gr["dst"].to(dev()), n)
    views = views_for(graph, dev())
    a = m(views, x, ef)
    assert a.shape == (e, 1) and torch.isfinite(a).all() and torch.equal(a, m(views, x, ef))
    if e == 2_500_000:   # the whole forward on round 2's streaming gate (variant 9): the plane form gives the same bits at full size
        try:
            ops.set_tuning(10, 1)   # the bf16x6 plane form (round 3) ...
            a6 = m(views, x, ef)
            ops.set_tuning(0, 9)    # ... and round 2's streaming gate: the same bits
            assert torch.equal(a6, m(views, x, ef))
        finally:
            ops.set_tuning(0, 0)
            ops.set_tuning(10, 0)
        assert _prob_diff(a, a6) < PROB_TOL / 10    # the fp16x3 default against them: one tenth of the parity bar
        del a6
    b = _model(_swap_roles(sd, hidden), hidden)(views.reversed(), x, ef)
    assert _prob_diff(a, b) < PROB_TOL
    perm = torch.randperm(e, generator=torch.Generator().manual_seed(3)).to(dev())
    d = m((graph[0][perm], graph[1][perm], n), x, ef[perm])
    assert _prob_diff(a[perm], d) < PROB_TOL
    del a, b, d
    torch.cuda.empty_cache()


def test_kernel_called_through_a_binding_built_from_the_header_alone():
    """The C ABI as a maintainer of the reference would bind it: ctypes prototypes parsed out of include/gnnome_hip.h
    (tests/header_binding.py), raw device pointers, torch only as the allocator."""
    import ctypes

    import header_binding
    from gnnome_amd import _lib
    lib = header_binding.bind(_lib.LIB_PATH, _lib.HEADER_PATH)
    g = torch.Generator().manual_seed(4)
    x, W1, b1 = torch.randn(999, 2, generator=g), torch.randn(16, 2, generator=g), torch.randn(16, generator=g)
    W2, b2 = torch.randn(128, 16, generator=g), torch.randn(128, generator=g)
    d = [t.to(dev()).contiguous() for t in (x, W1, b1, W2, b2)]
    out = torch.empty(999, 128, device=dev())
    p = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    rc = lib.gnnome_encode_f32(p(d[0]), 999, 2, None, p(d[1]), p(d[2]), 16, p(d[3]), p(d[4]), 128, p(out), stream)
    assert rc == 0, lib.gnnome_last_error()
    want = cpu_ops.encode(x.double(), W1.double(), b1.double(), W2.double(), b2.double())
    _assert_close(out, want)
    assert lib.gnnome_encode_f32(p(d[0]), 999, 2, None, p(d[1]), p(d[2]), 16, p(d[3]), p(d[4]), 96, p(out), stream) == -1
    assert b"hidden=96" in lib.gnnome_last_error()


def test_counter_synchronised_kernels_soak():
    """tools/gate_soak.py: a few hundred launches of the edge-tile kernels at random sizes and in every mode, each checked
    against the barrier-synchronised / tile kernels; bounded, so that a lost hand-over fails the test instead of hanging it."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, os.path.join(root, "tools", "gate_soak.py"), "3", "300"], capture_output=True, text=True,
                         timeout=300)
    assert res.returncode == 0 and "soak ok" in res.stdout, res.stdout[-500:] + res.stderr[-500:]
