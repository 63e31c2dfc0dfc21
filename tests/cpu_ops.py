"""CHECKER backend: torch-CPU statement of each C-ABI entry point's contract (include/gnnome_hip.h).

Test infrastructure only.  It lets the host logic in gnnome_amd/engine.py and gnnome_amd/dist.py
(kernel sequence, bias folding, reversed-graph table swap, partition + halo exchange) be checked on
CPU ranks against the oracle, and gives the GPU tests a per-kernel expected value.
"""
import torch

NORM_AFFINE, NORM_LAYER = 0, 1


class CpuViews:
    def __init__(self, src, dst, num_nodes):
        src, dst = torch.as_tensor(src).long().cpu(), torch.as_tensor(dst).long().cpu()
        n, e = int(num_nodes), src.numel()
        self.num_nodes, self.num_edges, self.device, self.transposed = n, e, torch.device("cpu"), False
        order = torch.sort(dst, stable=True).indices
        self.srt_eid = order.int()
        self.srt_src, self.srt_dst = src[order].int(), dst[order].int()
        self.in_ptr = torch.searchsorted(dst[order].contiguous(), torch.arange(n + 1)).int()
        pos = torch.sort(src[order], stable=True).indices
        self.out_pos = pos.int()
        self.out_ptr = torch.searchsorted(src[order][pos].contiguous(), torch.arange(n + 1)).int()
        self.out_dst = dst[order][pos].int()

    def reversed(self):
        r = object.__new__(CpuViews)
        r.__dict__.update(self.__dict__)
        r.transposed = not self.transposed
        return r


GraphViews = CpuViews


def _norm(x, kind, scale, shift):
    if kind == NORM_AFFINE:
        return x * scale + shift
    mu = x.mean(1, keepdim=True)
    var = ((x - mu) ** 2).mean(1, keepdim=True)
    return (x - mu) * torch.rsqrt(var + 1e-5) * scale + shift


def encode(x, W1, b1, W2, b2, gather=None, rows=None):
    if gather is not None:
        x = x[gather.long()]
    return torch.relu(x @ W1.t() + b1) @ W2.t() + b2


def linear(A, W, bias, out=None):
    y = A @ W.t()
    if bias is not None:
        y = y + bias
    if out is None:
        return y
    out.copy_(y)
    return out


def edge_gate(e, B1h, B2h, views, W3, norm_kind, scale, shift, out=None, num_edges=None):
    E = e.shape[0] if num_edges is None else num_edges
    s, d = views.srt_src[:E].long(), views.srt_dst[:E].long()
    x = B1h[s] + B2h[d] + e[:E] @ W3.t()
    y = torch.relu(_norm(x, norm_kind, scale, shift)) + e[:E]
    out = e if out is None else out
    out[:E] = y
    return out


def node_aggregate(e, A1h, A2h, A3h, views, h_in, norm_kind, scale, shift, num_nodes_out=None):
    n = h_in.shape[0]
    n_out = n if num_nodes_out is None else num_nodes_out
    sig = torch.sigmoid(e)
    s, d = views.srt_src.long(), views.srt_dst.long()
    zeros = torch.zeros_like(h_in)
    nf = zeros.index_add(0, d, sig * A2h[s])
    df = zeros.index_add(0, d, sig)
    nb = zeros.index_add(0, s, sig * A3h[d])
    db = zeros.index_add(0, s, sig)
    v = A1h + nf / (df + 1e-6) + nb / (db + 1e-6)
    y = torch.relu(_norm(v, norm_kind, scale, shift)) + h_in
    out = torch.full_like(h_in, float("nan"))  # rows >= n_out are not written by the kernel
    out[:n_out] = y[:n_out]
    return out


def edge_score(e, Ps, Qd, views, W1e, W2, b2, W3, b3, logits, num_edges=None, scatter_to_edge_id=True):
    E = e.shape[0] if num_edges is None else num_edges
    s, d = views.srt_src[:E].long(), views.srt_dst[:E].long()
    z1 = torch.relu(Ps[s] + Qd[d] + e[:E] @ W1e.t())
    z2 = torch.relu(z1 @ W2.t() + b2)
    val = z2 @ W3 + b3
    if scatter_to_edge_id:
        logits[views.srt_eid[:E].long()] = val
    else:
        logits[:E] = val
    return logits


def gather_rows(table, idx, out=None):
    y = table[idx.long()]
    if out is None:
        return y
    out.copy_(y)
    return out
