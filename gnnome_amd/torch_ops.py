"""`torch.ops.gnnome_hip.*`: the library's inference entry points as PyTorch dispatcher operators - a COMPILED extension.

The drop-in boundary of this package is the C ABI (include/gnnome_hip.h); gnnome_amd.ops binds it with ctypes for the model path.
SURVEY.md 8(b) / north_star also ask for the kernels to be reachable the way a PyTorch-ROCm extension's are - through the
dispatcher, with schemas, device checks and shape inference for tracing (torch.compile / FakeTensor) - so that a maintainer can call

    P      = torch.ops.gnnome_hip.linear(h, Wcat, bcat)
    e_new  = torch.ops.gnnome_hip.edge_gate(e, B1h, B2h, srt_src, srt_dst, W3, scale, shift, norm_kind)
    h_new  = torch.ops.gnnome_hip.node_aggregate(e_new, A1h, A2h, A3h, in_ptr, srt_src, out_ptr, out_pos, out_dst, h, scale, shift, norm_kind)
    logits = torch.ops.gnnome_hip.edge_score(e, Ps, Qd, srt_src, srt_dst, srt_eid, W1e, W2, b2, W3, b3)

from their own module code (or from C++ / TorchScript: the operators live in the dispatcher, not in Python).  They are defined and
implemented in gnnome_amd/csrc/torch_ext.cpp (TORCH_LIBRARY + TORCH_LIBRARY_IMPL under the CUDA key = HIP on ROCm, and Meta), built as
gnnome_amd/lib/libgnnome_torch.so next to libgnnome_hip.so, whose C entries they call on the caller's current stream.  On any other device
the dispatcher raises - there is no CPU kernel.  Inference operators (no autograd formula: the training step is one autograd.Function,
gnnome_amd/train.py).  Rounds 2-5 registered the same schemas from Python over the ctypes binding; round 6 compiled them.

Import this module to load the extension (gnnome_amd/__init__.py does not: plain use of the model needs no dispatcher entries).  A missing
or stale build raises here - nothing falls back to Python.
"""
import os

import torch

from . import _lib

EXT_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "libgnnome_torch.so")


def load():
    """Load libgnnome_hip.so (ABI-checked), then the compiled operator library bound to it."""
    _lib.load()
    if not os.path.isfile(EXT_PATH):
        raise _lib.GnnomeHipError(f"{EXT_PATH} not found - the PyTorch extension is not built. Run `make -C gnnome_amd/csrc torch_ext` "
                                  "(or `python -c 'import __graft_entry__ as g; g.build()'`).")
    torch.ops.load_library(EXT_PATH)
    got = torch.ops.gnnome_hip.abi_version()
    if got != _lib.ABI_VERSION:
        raise _lib.GnnomeHipError(f"libgnnome_torch.so is bound to a libgnnome_hip.so of ABI {got}, the binding expects {_lib.ABI_VERSION}; rebuild")


load()
