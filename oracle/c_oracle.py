"""ctypes front end of oracle/symgated_oracle.c (checker only; see that file's header)."""
import ctypes
import os
import subprocess

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))


def _lib(precision):
    path = os.path.join(_HERE, "_build", f"liboracle_{precision}.so")
    if not os.path.isfile(path):
        subprocess.check_call(["make", "-C", _HERE])
    return ctypes.CDLL(path)


def pack_state_dict(sd, dtype):
    """Flatten a reference-format state_dict in key order, skipping the integer num_batches_tracked entries."""
    parts = [v.detach().cpu().numpy().astype(dtype).ravel() for k, v in sd.items() if not k.endswith("num_batches_tracked")]
    return np.ascontiguousarray(np.concatenate(parts))


def forward(sd, src, dst, num_nodes, x, e, normalization="batch", precision="f64"):
    """logits[E,1] of the C restatement for a reference-format state_dict."""
    dtype = np.float64 if precision == "f64" else np.float32
    H = sd["linear2_node.weight"].shape[0]
    M, Fn = sd["linear1_node.weight"].shape
    Fe = sd["linear1_edge.weight"].shape[1]
    L = 1 + max(int(k.split(".")[2]) for k in sd if k.startswith("gnn.convs."))
    hs = sd["predictor.W1.weight"].shape[0]
    params = pack_state_dict(sd, dtype)
    src_a = np.ascontiguousarray(torch.as_tensor(src).cpu().numpy().astype(np.int32))
    dst_a = np.ascontiguousarray(torch.as_tensor(dst).cpu().numpy().astype(np.int32))
    xa = np.ascontiguousarray(x.cpu().numpy().astype(dtype))
    ea = np.ascontiguousarray(e.cpu().numpy().astype(dtype))
    E = src_a.shape[0]
    out = np.zeros(E, dtype=dtype)
    ptr = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    rc = _lib(precision).gnnome_oracle_forward(int(num_nodes), E, Fn, Fe, M, H, L, hs, 0 if normalization == "batch" else 1,
                                                ptr(src_a), ptr(dst_a), ptr(xa), ptr(ea), ptr(params), ptr(out))
    if rc != 0:
        raise MemoryError("gnnome_oracle_forward failed")
    return torch.from_numpy(out.astype(np.float64 if precision == "f64" else np.float32)).unsqueeze(1)
