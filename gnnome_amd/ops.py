"""Thin torch-tensor front end of the C ABI: pointers, sizes and the current HIP stream go in,
nothing else.  torch is used for device memory and stream ownership only."""
import ctypes

import torch

from . import _lib
from ._lib import NORM_AFFINE, NORM_LAYER  # noqa: F401


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr() if t is not None and t.numel() > 0 else None)


def _stream(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _f32(t, name):
    if t.dtype != torch.float32 or not t.is_cuda:
        raise TypeError(f"{name}: expected a CUDA float32 tensor, got {t.dtype} on {t.device}")
    return t


def _i32(t, name):
    if t.dtype != torch.int32 or not t.is_cuda or not t.is_contiguous():
        raise TypeError(f"{name}: expected a contiguous CUDA int32 tensor, got {t.dtype} on {t.device}")
    return t


def _rows(t, name):
    """2-D float32 with unit column stride -> (tensor, row stride in elements)."""
    _f32(t, name)
    if t.dim() != 2 or (t.shape[1] > 1 and t.stride(1) != 1):
        raise ValueError(f"{name}: expected a 2-D tensor with contiguous rows, got shape {tuple(t.shape)} strides {t.stride()}")
    return t, (t.stride(0) if t.shape[0] > 1 else max(t.stride(0), t.shape[1]))


class GraphViews:
    """In-edge / out-edge orderings of one edge list (see include/gnnome_hip.h, "graph views")."""

    __slots__ = ("num_nodes", "num_edges", "in_ptr", "srt_src", "srt_dst", "srt_eid", "out_ptr", "out_pos", "out_dst", "device",
                 "transposed", "__weakref__")

    def __init__(self, src, dst, num_nodes):
        lib = _lib.load()
        _i32(src, "src"), _i32(dst, "dst")
        dev = src.device
        n, e = int(num_nodes), int(src.numel())
        if dst.numel() != e:
            raise ValueError("src and dst differ in length")
        if e > 0:
            lo = int(torch.minimum(src.min(), dst.min()))
            hi = int(torch.maximum(src.max(), dst.max()))
            if lo < 0 or hi >= n:
                raise IndexError(f"edge endpoint out of range [0,{n}): min {lo}, max {hi}")
        self.num_nodes, self.num_edges, self.device, self.transposed = n, e, dev, False
        mk = lambda k: torch.empty(k, dtype=torch.int32, device=dev)  # noqa: E731
        self.in_ptr, self.out_ptr = mk(n + 1), mk(n + 1)
        self.srt_src, self.srt_dst, self.srt_eid, self.out_pos, self.out_dst = mk(e), mk(e), mk(e), mk(e), mk(e)
        need = ctypes.c_size_t(0)
        with torch.cuda.device(dev):
            _lib.check(lib.gnnome_graph_views_workspace_bytes(n, e, ctypes.byref(need)), "graph_views_workspace_bytes")
            ws = torch.empty(max(int(need.value), 1), dtype=torch.uint8, device=dev)
            _lib.check(lib.gnnome_build_graph_views(_ptr(src), _ptr(dst), n, e, _ptr(self.in_ptr), _ptr(self.srt_src),
                                                    _ptr(self.srt_dst), _ptr(self.srt_eid), _ptr(self.out_ptr),
                                                    _ptr(self.out_pos), _ptr(self.out_dst), _ptr(ws), ws.numel(),
                                                    _stream(dev)),
                       "build_graph_views")
            ws.record_stream(torch.cuda.current_stream(dev))

    def reversed(self):
        """Views of dgl.reverse(g, copy_ndata=True, copy_edata=True) - endpoints swapped, edge ids and edge
        storage order kept (gated_gcn_full.py:99, train.py:165).  Free: the same arrays, with the roles of
        the contiguous (by-dst) and permuted (by-src) runs exchanged by the caller (`transposed`)."""
        r = object.__new__(GraphViews)
        for k in GraphViews.__slots__:
            if k != "__weakref__":
                setattr(r, k, getattr(self, k))
        r.transposed = not self.transposed
        return r


def set_tuning(key, value):
    """Select a kernel variant for A/B measurements (see gnnome_set_tuning in include/gnnome_hip.h)."""
    _lib.check(_lib.load().gnnome_set_tuning(int(key), int(value)), "set_tuning")


def encode(x, W1, b1, W2, b2, gather=None, rows=None):
    lib = _lib.load()
    x, _ = _rows(x.contiguous(), "encode.in")
    hidden, hidden_ne = W2.shape[0], W1.shape[0]
    rows = int(x.shape[0] if rows is None else rows)
    out = torch.empty((rows, hidden), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(lib.gnnome_encode_f32(_ptr(x), rows, x.shape[1], _ptr(gather), _ptr(W1), _ptr(b1), hidden_ne, _ptr(W2),
                                         _ptr(b2), hidden, _ptr(out), _stream(x.device)), "encode_f32")
    return out


def linear(A, W, bias, out=None):
    """out[M,Nout] = A @ W.T + bias on the fp32 matrix cores.  A, W, out may be row-strided views."""
    lib = _lib.load()
    A, lda = _rows(A, "linear.A")
    W, ldw = _rows(W, "linear.W")
    M, K = A.shape
    Nout = W.shape[0]
    if out is None:
        out = torch.empty((M, Nout), dtype=torch.float32, device=A.device)
    out, ldc = _rows(out, "linear.out")
    with torch.cuda.device(A.device):
        _lib.check(lib.gnnome_linear_f32(_ptr(A), M, K, lda, _ptr(W), ldw, _ptr(bias), Nout, _ptr(out), ldc,
                                         _stream(A.device)), "linear_f32")
    return out


def edge_gate(e, B1h, B2h, views, W3, norm_kind, scale, shift, out=None, num_edges=None):
    lib = _lib.load()
    e, _ = _rows(e, "edge_gate.e")
    B1h, ldn = _rows(B1h, "edge_gate.B1h")
    B2h, ldn2 = _rows(B2h, "edge_gate.B2h")
    assert ldn == ldn2 and e.is_contiguous()
    W3, ldw = _rows(W3, "edge_gate.W3")
    out = e if out is None else out
    E = int(e.shape[0] if num_edges is None else num_edges)
    with torch.cuda.device(e.device):
        _lib.check(lib.gnnome_edge_gate_f32(_ptr(e), _ptr(out), E, e.shape[1], _ptr(B1h), _ptr(B2h), ldn,
                                            _ptr(views.srt_src), _ptr(views.srt_dst), _ptr(W3), ldw, norm_kind, _ptr(scale),
                                            _ptr(shift), _stream(e.device)), "edge_gate_f32")
    return out


def node_aggregate(e, A1h, A2h, A3h, views, h_in, norm_kind, scale, shift, num_nodes_out=None):
    lib = _lib.load()
    A1h, ldn = _rows(A1h, "node_aggregate.A1h")
    A2h, l2 = _rows(A2h, "node_aggregate.A2h")
    A3h, l3 = _rows(A3h, "node_aggregate.A3h")
    h_in, ldh = _rows(h_in, "node_aggregate.h_in")
    assert ldn == l2 == l3
    hidden = h_in.shape[1]
    n_out = int(h_in.shape[0] if num_nodes_out is None else num_nodes_out)
    h_out = torch.empty((h_in.shape[0], hidden), dtype=torch.float32, device=h_in.device)
    with torch.cuda.device(h_in.device):
        _lib.check(lib.gnnome_node_aggregate_f32(_ptr(e), hidden, n_out, _ptr(A1h), _ptr(A2h), _ptr(A3h), ldn,
                                                 _ptr(views.in_ptr), _ptr(views.srt_src), _ptr(views.out_ptr),
                                                 _ptr(views.out_pos), _ptr(views.out_dst), _ptr(h_in), ldh, _ptr(h_out),
                                                 norm_kind, _ptr(scale), _ptr(shift), _stream(h_in.device)),
                   "node_aggregate_f32")
    return h_out


def edge_score(e, Ps, Qd, views, W1e, W2, b2, W3, b3, logits, num_edges=None, scatter_to_edge_id=True):
    lib = _lib.load()
    e, _ = _rows(e, "edge_score.e")
    Ps, ldn = _rows(Ps, "edge_score.Ps")
    Qd, ldn2 = _rows(Qd, "edge_score.Qd")
    assert ldn == ldn2 and e.is_contiguous()
    W1e, ldw1 = _rows(W1e, "edge_score.W1e")
    E = int(e.shape[0] if num_edges is None else num_edges)
    eid = views.srt_eid if scatter_to_edge_id else None
    with torch.cuda.device(e.device):
        _lib.check(lib.gnnome_edge_score_f32(_ptr(e), E, e.shape[1], W2.shape[1], _ptr(Ps), _ptr(Qd), ldn,
                                             _ptr(views.srt_src), _ptr(views.srt_dst), _ptr(eid), _ptr(W1e), ldw1,
                                             _ptr(W2), _ptr(b2), _ptr(W3), _ptr(b3), _ptr(logits), _stream(e.device)),
                   "edge_score_f32")
    return logits


def gather_rows(table, idx, out=None):
    lib = _lib.load()
    table, ld_in = _rows(table, "gather_rows.in")
    _i32(idx, "gather_rows.idx")
    rows, width = int(idx.numel()), table.shape[1]
    if out is None:
        out = torch.empty((rows, width), dtype=torch.float32, device=table.device)
    out, ld_out = _rows(out, "gather_rows.out")
    with torch.cuda.device(table.device):
        _lib.check(lib.gnnome_gather_rows_f32(_ptr(table), ld_in, _ptr(idx), rows, width, _ptr(out), ld_out,
                                              _stream(table.device)), "gather_rows_f32")
    return out
