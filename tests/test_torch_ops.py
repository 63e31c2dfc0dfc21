"""torch.ops.gnnome_hip.* (gnnome_amd/torch_ops.py): the C ABI's inference entry points as dispatcher operators -
schemas and shape inference here, the kernels behind them on the GPU."""
import pytest
import torch

import cpu_ops


@pytest.fixture(autouse=True, scope="module")
def _operators():
    """Load the compiled extension (gnnome_amd/lib/libgnnome_torch.so) when the first test of this module runs - a missing or stale build fails
    these tests, not the collection of the whole suite."""
    import gnnome_amd.torch_ops  # noqa: F401


def test_operators_are_registered_with_schemas_and_trace_on_meta():
    names = ("build_graph_views", "encode", "linear", "linear_ref", "edge_gate", "node_aggregate", "edge_score")
    for n in names:
        assert hasattr(torch.ops.gnnome_hip, n)
    assert "Tensor? bias=None" in str(torch.ops.gnnome_hip.linear.default._schema)
    # shape inference without a device (what FakeTensor tracing uses)
    n, e, H = 50, 400, 64
    m = lambda *s: torch.empty(s, device="meta")  # noqa: E731
    views = torch.ops.gnnome_hip.build_graph_views(torch.empty(e, dtype=torch.int32, device="meta"), torch.empty(e, dtype=torch.int32, device="meta"), n)
    assert [tuple(v.shape) for v in views] == [(n + 1,), (e,), (e,), (e,), (n + 1,), (e,), (e,)]
    assert torch.ops.gnnome_hip.linear(m(n, H), m(5 * H, H), m(5 * H)).shape == (n, 5 * H)
    assert torch.ops.gnnome_hip.edge_gate(m(e, H), m(n, H), m(n, H), views[1], views[2], m(H, H), m(H), m(H)).shape == (e, H)
    assert torch.ops.gnnome_hip.edge_score(m(e, H), m(n, 64), m(n, 64), views[1], views[2], views[3], m(64, H), m(32, 64), m(32), m(32), m(1)).shape == (e,)
    # there is no CPU kernel behind the operators
    with pytest.raises((NotImplementedError, RuntimeError)):
        torch.ops.gnnome_hip.linear(torch.zeros(4, 64), torch.zeros(8, 64), None)


@pytest.mark.gpu
def test_operators_run_the_hip_kernels():
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(0)
    n, e, H = 300, 2500, 128
    src, dst = torch.randint(0, n, (e,), generator=g).int(), torch.randint(0, n, (e,), generator=g).int()
    cv = cpu_ops.CpuViews(src, dst, n)
    in_ptr, ss, sd, eid, out_ptr, out_pos, out_dst = torch.ops.gnnome_hip.build_graph_views(src.to(dev), dst.to(dev), n)
    assert torch.equal(eid.cpu(), cv.srt_eid) and torch.equal(out_pos.cpu(), cv.out_pos)
    h, ee = torch.randn(n, H, generator=g), 3 * torch.randn(e, H, generator=g)
    Wc, bc = torch.randn(5 * H, H, generator=g) / H ** 0.5, torch.randn(5 * H, generator=g)
    W3 = torch.randn(H, H, generator=g) / H ** 0.5
    sc, sh = 0.5 + torch.rand(H, generator=g), torch.randn(H, generator=g)
    P = torch.ops.gnnome_hip.linear(h.to(dev), Wc.to(dev), bc.to(dev))
    want_P = cpu_ops.linear(h.double(), Wc.double(), bc.double())
    assert (P.cpu().double() - want_P).abs().max() < 1e-4
    A1, A2, A3, B1, B2 = (P[:, i * H:(i + 1) * H] for i in range(5))
    e_in = ee.to(dev)
    e_new = torch.ops.gnnome_hip.edge_gate(e_in, B1, B2, ss, sd, W3.to(dev), sc.to(dev), sh.to(dev), 0)
    assert torch.equal(e_in.cpu(), ee)                       # functional: the input is not modified
    Pd = P.cpu().double()
    want_e = cpu_ops.edge_gate(ee.double().clone(), Pd[:, 3 * H:4 * H], Pd[:, 4 * H:], cv, W3.double(), 0, sc.double(), sh.double())
    assert (e_new.cpu().double() - want_e).abs().max() < 1e-3
    h_new = torch.ops.gnnome_hip.node_aggregate(e_new, A1, A2, A3, in_ptr, ss, out_ptr, out_pos, out_dst, h.to(dev), sc.to(dev), sh.to(dev), 0)
    want_h = cpu_ops.node_aggregate(want_e, Pd[:, :H], Pd[:, H:2 * H], Pd[:, 2 * H:3 * H], cv, h.double(), 0, sc.double(), sh.double())
    assert (h_new.cpu().double() - want_h).abs().max() < 1e-3


def test_operators_are_compiled_kernels_not_python():
    """The CUDA (= HIP) and Meta kernels are registered from csrc/torch_ext.cpp, and the extension is bound to the library the ctypes side loaded."""
    for n in ("build_graph_views", "encode", "linear", "linear_ref", "edge_gate", "node_aggregate", "edge_score"):
        dump = torch._C._dispatch_dump(f"gnnome_hip::{n}")
        assert "CUDA: registered at" in dump and "Meta: registered at" in dump and "torch_ext.cpp" in dump and "PythonModule" not in dump, dump
    from gnnome_amd import _lib
    assert torch.ops.gnnome_hip.abi_version() == _lib.ABI_VERSION


@pytest.mark.gpu
def test_operators_equal_the_ctypes_path_and_follow_the_current_stream():
    """Same C entries underneath: bit-equal to gnnome_amd.ops; strided column blocks are taken as they are; the launch goes to the caller's stream."""
    from gnnome_amd import ops
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(1)
    n, e, H, hs = 500, 4000, 128, 64
    src, dst = torch.randint(0, n, (e,), generator=g).int().to(dev), torch.randint(0, n, (e,), generator=g).int().to(dev)
    views = ops.GraphViews(src, dst, n)
    got = torch.ops.gnnome_hip.build_graph_views(src.long(), dst, n)   # any integer dtype comes in
    for a, b in zip(got, (views.in_ptr, views.srt_src, views.srt_dst, views.srt_eid, views.out_ptr, views.out_pos, views.out_dst)):
        assert torch.equal(a, b)
    with pytest.raises(IndexError):
        torch.ops.gnnome_hip.build_graph_views(src, dst, n - 50)
    h, ee = torch.randn(n, H, generator=g).to(dev), (3 * torch.randn(e, H, generator=g)).to(dev)
    Wc, bc = (torch.randn(5 * H, H, generator=g) / H ** 0.5).to(dev), torch.randn(5 * H, generator=g).to(dev)
    W3 = (torch.randn(H, H, generator=g) / H ** 0.5).to(dev)
    sc, sh = (0.5 + torch.rand(H, generator=g)).to(dev), torch.randn(H, generator=g).to(dev)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        P = torch.ops.gnnome_hip.linear(h, Wc, bc)
        A1, A2, A3, B1, B2 = (P[:, i * H:(i + 1) * H] for i in range(5))
        e_new = torch.ops.gnnome_hip.edge_gate(ee, B1, B2, views.srt_src, views.srt_dst, W3, sc, sh)
        h_new = torch.ops.gnnome_hip.node_aggregate(e_new, A1, A2, A3, views.in_ptr, views.srt_src, views.out_ptr, views.out_pos, views.out_dst, h, sc, sh)
    torch.cuda.current_stream().wait_stream(side)
    P2 = ops.linear(h, Wc, bc)
    assert torch.equal(P, P2)
    e2 = ops.edge_gate(ee.clone(), P2[:, 3 * H:4 * H], P2[:, 4 * H:], views, W3, 0, sc, sh)
    assert torch.equal(e_new, e2)
    assert torch.equal(h_new, ops.node_aggregate(e2, P2[:, :H], P2[:, H:2 * H], P2[:, 2 * H:3 * H], views, h, 0, sc, sh))
    assert torch.equal(torch.ops.gnnome_hip.linear_ref(h, Wc, bc), ops.linear_ref(h, Wc, bc))
    assert torch.equal(torch.ops.gnnome_hip.linear(h, Wc[:64], None), ops.linear(h, Wc[:64], None))     # a shape outside the planes route
    # encoder with and without the gather; the scorer
    x = torch.randn(e, 2, generator=g).to(dev)
    W1, b1 = torch.randn(16, 2, generator=g).to(dev), torch.randn(16, generator=g).to(dev)
    W2, b2 = (torch.randn(H, 16, generator=g) / 4).to(dev), torch.randn(H, generator=g).to(dev)
    assert torch.equal(torch.ops.gnnome_hip.encode(x, W1, b1, W2, b2), ops.encode(x, W1, b1, W2, b2))
    assert torch.equal(torch.ops.gnnome_hip.encode(x, W1, b1, W2, b2, views.srt_eid), ops.encode(x, W1, b1, W2, b2, gather=views.srt_eid, rows=e))
    PQ = torch.randn(n, 2 * hs, generator=g).to(dev)
    W1e = (torch.randn(hs, H, generator=g) / H ** 0.5).to(dev)
    V2, c2 = (torch.randn(32, hs, generator=g) / 8).to(dev), torch.randn(32, generator=g).to(dev)
    V3, c3 = torch.randn(32, generator=g).to(dev), torch.randn(1, generator=g).to(dev)
    got = torch.ops.gnnome_hip.edge_score(e_new, PQ[:, :hs], PQ[:, hs:], views.srt_src, views.srt_dst, views.srt_eid, W1e, V2, c2, V3, c3)
    want = ops.edge_score(e_new, PQ[:, :hs], PQ[:, hs:], views, W1e, V2, c2, V3, c3, torch.empty(e, device=dev))
    assert torch.equal(got, want)
    # what the operators refuse: the wrong dtype, a transposed matrix, tables of different projections
    with pytest.raises(RuntimeError):
        torch.ops.gnnome_hip.linear(h.double(), Wc, bc)
    with pytest.raises(RuntimeError):
        torch.ops.gnnome_hip.linear(h, Wc.t().contiguous().t(), bc)
    with pytest.raises(RuntimeError):
        torch.ops.gnnome_hip.edge_gate(ee, B1, B2.contiguous(), views.srt_src, views.srt_dst, W3, sc, sh)
    with pytest.raises(RuntimeError):
        torch.ops.gnnome_hip.edge_gate(ee, B1, B2, views.srt_src.long(), views.srt_dst, W3, sc, sh)
