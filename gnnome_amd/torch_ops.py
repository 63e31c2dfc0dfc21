"""`torch.ops.gnnome_hip.*`: the library's entry points as PyTorch dispatcher operators.

The drop-in boundary of this package is the C ABI (include/gnnome_hip.h); gnnome_amd.ops binds it with ctypes.  SURVEY.md
8(b) also asks for the kernels to be reachable the way a PyTorch extension's are - through the dispatcher, with schemas,
device checks and shape inference for tracing (torch.compile / FakeTensor) - so that a maintainer can call

    P      = torch.ops.gnnome_hip.linear(h, Wcat, bcat)
    e_new  = torch.ops.gnnome_hip.edge_gate(e, B1h, B2h, srt_src, srt_dst, W3, scale, shift, norm_kind)
    h_new  = torch.ops.gnnome_hip.node_aggregate(e_new, A1h, A2h, A3h, in_ptr, srt_src, out_ptr, out_pos, out_dst, h, scale, shift, norm_kind)
    logits = torch.ops.gnnome_hip.edge_score(e, Ps, Qd, srt_src, srt_dst, srt_eid, W1e, W2, b2, W3, b3)

from their own module code.  The operators are registered from Python (torch.library), dispatch key CUDA (= HIP on ROCm),
and forward to the same C ABI calls; on any other device the dispatcher raises - there is no CPU kernel.  Inference
operators (no autograd formula: the training step is one autograd.Function, gnnome_amd/train.py).

Import this module to register the operators (gnnome_amd/__init__.py does not import it: plain use of the model needs no
dispatcher entries).
"""
import torch
from torch.library import Library

from . import ops

_lib = Library("gnnome_hip", "DEF")


class _Views:
    """The slice of GraphViews a kernel reads, rebuilt from the operator's tensor arguments."""

    def __init__(self, **kw):
        self.transposed = False
        self.__dict__.update(kw)


_lib.define("build_graph_views(Tensor src, Tensor dst, int num_nodes) -> (Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor)")
_lib.define("encode(Tensor x, Tensor W1, Tensor b1, Tensor W2, Tensor b2, Tensor? gather=None) -> Tensor")
_lib.define("linear(Tensor A, Tensor W, Tensor? bias=None) -> Tensor")
_lib.define("linear_ref(Tensor A, Tensor W, Tensor? bias=None) -> Tensor")
_lib.define("edge_gate(Tensor e, Tensor B1h, Tensor B2h, Tensor srt_src, Tensor srt_dst, Tensor W3, Tensor scale, Tensor shift, "
            "int norm_kind=0) -> Tensor")
_lib.define("node_aggregate(Tensor e, Tensor A1h, Tensor A2h, Tensor A3h, Tensor in_ptr, Tensor srt_src, Tensor out_ptr, Tensor out_pos, "
            "Tensor out_dst, Tensor h_in, Tensor scale, Tensor shift, int norm_kind=0) -> Tensor")
_lib.define("edge_score(Tensor e, Tensor Ps, Tensor Qd, Tensor srt_src, Tensor srt_dst, Tensor srt_eid, Tensor W1e, Tensor W2, Tensor b2, "
            "Tensor W3, Tensor b3) -> Tensor")


def _build_graph_views(src, dst, num_nodes):
    v = ops.GraphViews(src.int().contiguous(), dst.int().contiguous(), num_nodes)
    return v.in_ptr, v.srt_src, v.srt_dst, v.srt_eid, v.out_ptr, v.out_pos, v.out_dst


def _encode(x, W1, b1, W2, b2, gather=None):
    return ops.encode(x, W1, b1, W2, b2, gather=gather, rows=None if gather is None else int(gather.numel()))


def _linear(A, W, bias=None):
    return ops.linear(A, W, bias)


def _linear_ref(A, W, bias=None):
    return ops.linear_ref(A, W, bias)


def _edge_gate(e, B1h, B2h, srt_src, srt_dst, W3, scale, shift, norm_kind=0):
    out = torch.empty_like(e)   # functional form: the dispatcher operator does not mutate its input
    ops.edge_gate(e.contiguous(), B1h, B2h, _Views(srt_src=srt_src, srt_dst=srt_dst), W3, norm_kind, scale, shift, out=out)
    return out


def _node_aggregate(e, A1h, A2h, A3h, in_ptr, srt_src, out_ptr, out_pos, out_dst, h_in, scale, shift, norm_kind=0):
    v = _Views(in_ptr=in_ptr, srt_src=srt_src, out_ptr=out_ptr, out_pos=out_pos, out_dst=out_dst)
    return ops.node_aggregate(e, A1h, A2h, A3h, v, h_in, norm_kind, scale, shift)


def _edge_score(e, Ps, Qd, srt_src, srt_dst, srt_eid, W1e, W2, b2, W3, b3):
    logits = torch.empty(e.shape[0], dtype=torch.float32, device=e.device)
    ops.edge_score(e, Ps, Qd, _Views(srt_src=srt_src, srt_dst=srt_dst, srt_eid=srt_eid), W1e, W2, b2, W3, b3, logits)
    return logits


for _name, _fn in (("build_graph_views", _build_graph_views), ("encode", _encode), ("linear", _linear), ("linear_ref", _linear_ref),
                   ("edge_gate", _edge_gate), ("node_aggregate", _node_aggregate), ("edge_score", _edge_score)):
    _lib.impl(_name, _fn, "CUDA")


# shape inference for tracing (FakeTensor / torch.compile): no kernel runs
def _meta_views(src, dst, num_nodes):
    e = src.shape[0]
    i32 = lambda n: torch.empty(n, dtype=torch.int32, device=src.device)  # noqa: E731
    return i32(num_nodes + 1), i32(e), i32(e), i32(e), i32(num_nodes + 1), i32(e), i32(e)


_lib.impl("build_graph_views", _meta_views, "Meta")
_lib.impl("encode", lambda x, W1, b1, W2, b2, gather=None: x.new_empty(((x.shape[0] if gather is None else gather.shape[0]), W2.shape[0])), "Meta")
_lib.impl("linear", lambda A, W, bias=None: A.new_empty((A.shape[0], W.shape[0])), "Meta")
_lib.impl("linear_ref", lambda A, W, bias=None: A.new_empty((A.shape[0], W.shape[0])), "Meta")
_lib.impl("edge_gate", lambda e, *a, **k: torch.empty_like(e), "Meta")
_lib.impl("node_aggregate", lambda e, A1h, A2h, A3h, in_ptr, srt_src, out_ptr, out_pos, out_dst, h_in, scale, shift, norm_kind=0:
          torch.empty_like(h_in), "Meta")
_lib.impl("edge_score", lambda e, *a: e.new_empty((e.shape[0],)), "Meta")
