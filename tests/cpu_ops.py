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
    w = (kind >> 8) or x.shape[1]      # GNNOME_NORM_LAYER_OVER(w): statistics over the first w channels (the rest are zero padding)
    mu = x[:, :w].mean(1, keepdim=True)
    var = ((x[:, :w] - mu) ** 2).mean(1, keepdim=True)
    return (x - mu) * torch.rsqrt(var + 1e-5) * scale + shift


def encode(x, W1, b1, W2, b2, gather=None, rows=None):
    if gather is not None:
        x = x[gather.long()]
    return torch.relu(x @ W1.t() + b1) @ W2.t() + b2


def linear(A, W, bias, out=None, accumulate=False, planes=None):   # planes: the HIP backend's prepared form of W, ignored here
    y = A @ W.t()
    if bias is not None:
        y = y + bias
    if out is None:
        return y
    if accumulate:
        out.add_(y)
    else:
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


def linear_ref(A, W, bias, out=None):
    """torch's own CPU nn.Linear IS the reference order (MKL: k-ascending fma chain from zero, then + bias)."""
    y = torch.nn.functional.linear(A, W, bias)
    if out is None:
        return y
    out.copy_(y)
    return out


def reference_order_supported(hidden, norm_kind, B1h=None):
    return hidden in (64, 128, 256) and norm_kind == NORM_AFFINE


def _fma(a, b, c):
    return (a.double() * b.double() + c.double()).to(a.dtype)   # fp32: exact product, one rounding (double rounding ~2^-29)


def edge_gate_ref(e, B1h, B2h, views, W3, b3, scale, shift, raw_edges=None, num_edges=None):
    if e is None:
        e_raw, (W1, b1, W2, b2) = raw_edges
        e = torch.nn.functional.linear(torch.relu(torch.nn.functional.linear(e_raw[views.srt_eid.long()], W1, b1)), W2, b2)
    E = e.shape[0] if num_edges is None else num_edges
    s, d = views.srt_src[:E].long(), views.srt_dst[:E].long()
    x = (B1h[s] + B2h[d]) + torch.nn.functional.linear(e[:E], W3, b3)
    e[:E] = torch.relu(_fma(x, scale, shift)) + e[:E]
    return e


def node_aggregate(e, A1h, A2h, A3h, views, h_in, norm_kind, scale, shift, num_nodes_out=None, node_range=None, out=None):
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
    if node_range is not None:   # gnnome_node_aggregate_range_f32: only these rows of the caller's `out`
        out[node_range[0]:node_range[1]] = y[node_range[0]:node_range[1]]
        return out
    out = torch.full_like(h_in, float("nan"))  # rows >= n_out are not written by the kernel
    out[:n_out] = y[:n_out]
    return out


def edge_score(e, Ps, Qd, views, W1e, W2, b2, W3, b3, logits, num_edges=None, scatter_to_edge_id=True, z1_out=None):
    E = e.shape[0] if num_edges is None else num_edges
    s, d = views.srt_src[:E].long(), views.srt_dst[:E].long()
    z1 = torch.relu(Ps[s] + Qd[d] + e[:E] @ W1e.t())
    if z1_out is not None:
        z1_out[:E] = z1
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


def scatter_add_rows(src, idx, out):
    assert idx.unique().numel() == idx.numel(), "scatter_add_rows needs distinct rows"
    out.index_add_(0, idx.long(), src)
    return out


# ---- training-step entries (include/gnnome_hip.h, "Training step") ---------------------------------

def edge_gate_raw(e, B1h, B2h, views, W3):
    return B1h[views.srt_src.long()] + B2h[views.srt_dst.long()] + e @ W3.t()


def edge_gate_raw_stats(e, B1h, B2h, views, W3, rows_stats=None):
    xe = edge_gate_raw(e, B1h, B2h, views, W3)
    mean, var = batch_stats(xe if rows_stats is None else xe[:rows_stats])
    return xe, mean, var


def node_aggregate_raw(e, A1h, A2h, A3h, views, mode, num_nodes, rows_alloc=None):
    s, d = views.srt_src.long(), views.srt_dst.long()
    n_tab, H = A2h.shape
    sig = torch.sigmoid(e)
    zeros = torch.zeros((n_tab, H), dtype=torch.float32)
    rows = num_nodes if rows_alloc is None else max(rows_alloc, num_nodes)

    def cut(t):  # the kernel computes rows < num_nodes only
        out = torch.zeros((rows, H), dtype=torch.float32)
        out[:num_nodes] = t[:num_nodes]
        return out

    sum_in = zeros.index_add(0, d, sig * A2h[s])
    sum_out = zeros.index_add(0, s, sig * A3h[d])
    if mode == 2:
        return cut(sum_in), cut(sum_out)
    rdf = 1.0 / (zeros.index_add(0, d, sig) + 1e-6)
    rdb = 1.0 / (zeros.index_add(0, s, sig) + 1e-6)
    fwd, bwd = sum_in * rdf, sum_out * rdb
    return cut(A1h + fwd + bwd), cut(fwd), cut(rdf), cut(bwd), cut(rdb)


def colsum2(x, y=None, center=None):
    xc = x if center is None else x - center
    yc = xc if y is None else y
    return xc.sum(0), (xc * yc).sum(0)


def batch_stats(x):
    return x.mean(0), x.var(0, unbiased=False)


def batch_moments(x):
    c = x[0]
    return (x - c).sum(0), ((x - c) ** 2).sum(0), c, x.shape[0]


def edge_gate_raw_moments(e, B1h, B2h, views, W3, storage=torch.float32):
    xe = edge_gate_raw(e, B1h, B2h, views, W3).to(storage)     # bf16 storage: rounded to nearest even, statistics of the rounded rows
    return xe, batch_moments(xe.float())


def can_fuse_gate_moments(e, B1h, B2h, storage=None):
    return e.shape[0] > 0


def can_two_pass_gate(e, B1h, B2h, storage=None):
    return e.shape[0] > 0 and e.shape[1] == 128


def edge_gate_moments_only(e, B1h, B2h, views, W3, storage=torch.float32):
    return edge_gate_raw_moments(e, B1h, B2h, views, W3, storage)[1]


def edge_gate_bn(e, B1h, B2h, views, W3, scale, shift, storage=torch.float32):
    xe = edge_gate_raw(e, B1h, B2h, views, W3).to(storage)
    return bn_relu_res(xe, scale, shift, e), xe


def bn_train_finish(moments, weight, bias, running_mean, running_var, num_batches_tracked, momentum, eps, updates):
    d1, d2, c, rows = moments
    m1 = d1 / rows
    mean, var = c + m1, (d2 / rows - m1 * m1).clamp_min(0.0)
    rstd = 1.0 / torch.sqrt(var + eps)
    scale = weight * rstd
    shift = bias - mean * scale
    if running_mean is not None:
        unbiased = var * (rows / max(rows - 1, 1))
        for _ in range(updates):
            running_mean.mul_(1 - momentum).add_(mean, alpha=momentum)
            running_var.mul_(1 - momentum).add_(unbiased, alpha=momentum)
    if num_batches_tracked is not None:
        num_batches_tracked += updates
    return mean, rstd, scale, shift


def bn_relu_res(x, scale, shift, res, out=None):
    y = torch.relu(x.float() * scale + shift) + res
    if out is None:
        return y
    out.copy_(y)
    return out


def bn_bwd_stats(dy, x, scale, shift, mean):
    g = dy * (x * scale + shift > 0)
    return g.sum(0), (g * (x - mean)).sum(0)


def bn_bwd_apply(dy, x, scale, shift, a, c1, c2, mean, rstd, out=None):
    dx = a * (dy * (x * scale + shift > 0) - c1 - (x - mean) * rstd * c2)
    if out is None:
        return dx
    out.copy_(dx)
    return out


def bn_bwd_apply_tables(dy, x, scale, shift, a, c1, c2, mean, rstd, rdf, hf, rdb, hb, out=None):
    dx = bn_bwd_apply(dy, x, scale, shift, a, c1, c2, mean, rstd, out=out)
    tf, tb = dx * rdf, dx * rdb
    return dx, tf, tf * hf, tb, tb * hb


def _ln_parts(x, width=None):
    w = width or x.shape[1]
    mu = x[:, :w].mean(1, keepdim=True)
    rstd = torch.rsqrt(((x[:, :w] - mu) ** 2).mean(1, keepdim=True) + 1e-5)
    xh = (x - mu) * rstd
    xh[:, w:] = 0
    return xh, rstd


def ln_relu_res(x, gamma, beta, res, out=None, width=None):
    y = torch.relu(_ln_parts(x, width)[0] * gamma + beta) + res
    if out is None:
        return y
    out.copy_(y)
    return out


def ln_bwd(dy, x, gamma, beta, out=None, width=None):
    w = width or x.shape[1]
    xh, rstd = _ln_parts(x, width)
    dm = dy * (xh * gamma + beta > 0)
    g = dm * gamma
    dx = rstd * (g - g.sum(1, keepdim=True) / w - xh * (g * xh).sum(1, keepdim=True) / w)
    dx[:, w:] = 0
    if out is not None:
        out.copy_(dx)
        dx = out
    return dx, (dm * xh).sum(0), dm.sum(0)


def mul23(a, b, c):
    return a * b, a * b * c


def add(a, b, out=None):
    if out is None:
        return a + b
    out.copy_(a + b)
    return out


def relu_bwd(dy, y):
    return dy * (y > 0)


def segment_sum(X, ptr, pos, num_nodes, out=None):
    ptr = ptr.long()
    counts = ptr[1:num_nodes + 1] - ptr[:num_nodes]
    seg = torch.repeat_interleave(torch.arange(num_nodes), counts)
    q = torch.arange(int(ptr[0]), int(ptr[num_nodes]))
    rows = X[q if pos is None else pos.long()[q]]
    res = torch.zeros((num_nodes, X.shape[1]), dtype=torch.float32).index_add(0, seg, rows)
    if out is None:
        return res
    out.copy_(res)
    return out


def wgrad(A, B, out=None):
    res = A.float().t() @ B
    if out is None:
        return res
    out.copy_(res)
    return out


def wgrad_blocks(blocks, B, colsum=True):
    A = torch.cat(list(blocks), 1)
    return A.t() @ B, (A.sum(0) if colsum else None)


def linear_blocks(blocks, W, out, accumulate=False):
    res = torch.cat(list(blocks), 1) @ W.t()
    if accumulate:
        out.add_(res)
    else:
        out.copy_(res)
    return out


def can_use_blocks(blocks):
    return True


def score_tail_bwd(z1, dscore, views, W2, b2, W3):
    E = z1.shape[0]
    dl = dscore[views.srt_eid[:E].long()]
    z2 = torch.relu(z1 @ W2.t() + b2)
    u = dl[:, None] * z2
    dz2 = dl[:, None] * W3[None, :] * (z2 > 0)
    dz1 = (dz2 @ W2) * (z1 > 0)
    return dz1, dz2, u


def agg_edge_bwd(e, Tf, Uf, Tb, Ub, A2h, A3h, views, de):
    s, d = views.srt_src.long(), views.srt_dst.long()
    sig = torch.sigmoid(e)
    de.add_(sig * (1 - sig) * (Tf[d] * A2h[s] - Uf[d] + Tb[s] * A3h[d] - Ub[s]))
    return de


def encode_hidden(x, W1, b1, gather=None, rows=None):
    if gather is not None:
        x = x[gather.long()]
    return torch.relu(x @ W1.t() + b1)


def agg_edge_bwd_stats(e, Tf, Uf, Tb, Ub, A2h, A3h, views, de, xe, scale, shift, mean):
    agg_edge_bwd(e, Tf, Uf, Tb, Ub, A2h, A3h, views, de)
    s1, s2 = bn_bwd_stats(de, xe.float(), scale, shift, mean)
    return de, s1, s2


def agg_bwd_fused(e, Tf, Uf, Tb, Ub, A2h, A3h, views, de, xe, scale, shift, mean, num_nodes):
    """ops.agg_bwd_fused: node_aggregate_raw mode 2 (tables Tb at src, Tf at dst) and agg_edge_bwd_stats as one op."""
    sum_in, sum_out = node_aggregate_raw(e, None, Tb, Tf, views, 2, num_nodes)
    _, s1, s2 = agg_edge_bwd_stats(e, Tf, Uf, Tb, Ub, A2h, A3h, views, de, xe, scale, shift, mean)
    return sum_in, sum_out, de, s1, s2


def can_fuse_bn_bwd_dgrad(de, W, xe=None):
    return de.shape[0] > 0


def bn_bwd_dgrad(de, xe, scale, shift, a, c1, c2, mean, rstd, Wt, rows_once=None):
    once = de.shape[0] if rows_once is None else int(rows_once)
    dxe = torch.empty_like(de)
    dxe[:once] = bn_bwd_apply(de[:once], xe[:once].float(), scale, shift, a, c1, c2, mean, rstd)
    if once < de.shape[0]:   # replicas of rows another rank owns: no mean-subtraction terms
        zero = torch.zeros_like(c1)
        dxe[once:] = bn_bwd_apply(de[once:], xe[once:].float(), scale, shift, a, zero, zero, mean, rstd)
    de.add_(dxe @ Wt.t())
    return dxe.to(xe.dtype)     # stored the way xe is; the product above used the unrounded values


def segment_sum2(X, views, num_nodes, out_in=None, out_out=None):
    X = X.float()
    a = segment_sum(X, views.in_ptr, None, num_nodes, out=out_in)
    b = segment_sum(X, views.out_ptr, views.out_pos, num_nodes, out=out_out)
    return a, b


def stream_schedule(views, chunks, rps=16, slots=62):
    """The streaming aggregation's schedule (include/gnnome_hip.h, gnnome_build_stream_schedule) restated in plain Python: chunk bounds,
    near / far rows, interval colouring of the nodes' live ranges with `slots` colours (lowest free first), step descriptors.
    -> dict(chunk_node, chunk_steps, steps {chunk: [(row, node, word2, pending?)]}, edge_meta, pending (set of nodes), far, overflow)."""
    import numpy as np
    FAR, UNALLOC, OVERFLOW, LATE_DISTANCE = 0xFF, 0xFF, 0xFE, 3
    MIDDLE, FIRST, LAST_FINAL, LAST_PENDING = 0, 1, 2, 3
    D_LASTSTEP, D_FIRST, D_LAST, D_HASSLOT = 1 << 5, 1 << 6, 1 << 7, 1 << 8
    n, e = views.num_nodes, views.num_edges
    in_ptr, srt_src = views.in_ptr.cpu().numpy().astype(np.int64), views.srt_src.cpu().numpy().astype(np.int64)
    out_ptr, out_pos, out_dst = (t.cpu().numpy().astype(np.int64) for t in (views.out_ptr, views.out_pos, views.out_dst))
    chunk_node = np.zeros(chunks + 1, dtype=np.int64)
    for c in range(1, chunks):
        chunk_node[c] = np.searchsorted(in_ptr[:n], e * c // chunks, side="left")
    chunk_node[chunks] = n
    meta = np.zeros(e, dtype=np.int64)
    last_near = np.arange(n)
    pending, far = set(), 0
    chunk_of = np.zeros(n, dtype=np.int64)
    for c in range(chunks):
        chunk_of[chunk_node[c]:chunk_node[c + 1]] = c   # (empty chunks write nothing: a node belongs to the last chunk that starts at or before it)
    for s in range(n):
        n0, n1 = chunk_node[chunk_of[s]], chunk_node[chunk_of[s] + 1]
        k0, k1 = out_ptr[s], out_ptr[s + 1]
        for k in range(k0, k1):
            d = out_dst[k]
            if n0 <= d < n1 and not (k > k0 and out_dst[k - 1] == d):
                last_near[s] = max(last_near[s], d)
            else:
                meta[out_pos[k]] = FAR
                far += 1
                pending.add(s)
        if 0 < last_near[s] - s < LATE_DISTANCE:   # its closing row would follow its in-run end too closely for the stream's prefetch distance
            pending.add(s)
    slot_of = np.full(n, UNALLOC, dtype=np.int64)
    steps, chunk_steps, overflow = {}, np.zeros(chunks, dtype=np.int64), 0
    for c in range(chunks):
        free = list(range(slots))   # kept sorted: lowest free slot first
        out = []
        for t in range(chunk_node[c], chunk_node[c + 1]):
            ib, ie = in_ptr[t], in_ptr[t + 1]
            free_after = []
            for p in range(ib, ie):
                if meta[p] == FAR:
                    continue
                s = srt_src[p]
                so = slot_of[s]
                if so == OVERFLOW:
                    meta[p] = FAR
                    far += 1
                    continue
                state = MIDDLE
                if so == UNALLOC:
                    if not free:
                        slot_of[s] = OVERFLOW
                        meta[p] = FAR
                        far += 1
                        pending.add(s)
                        overflow += 1
                        continue
                    so = free.pop(0)
                    slot_of[s] = so
                    state = FIRST
                if last_near[s] == t and s != t:
                    state = LAST_PENDING if s in pending else LAST_FINAL
                    free_after.append(so)
                meta[p] = (state << 6) | so
            so, info, last_ev = slot_of[t], 0, last_near[t] == t
            if so == OVERFLOW:
                info = D_FIRST | D_LAST
            elif so == UNALLOC:
                if last_ev:
                    info = D_FIRST | D_LAST
                elif not free:
                    slot_of[t] = OVERFLOW
                    pending.add(t)
                    overflow += 1
                    info = D_FIRST | D_LAST
                else:
                    so = free.pop(0)
                    slot_of[t] = so
                    info = D_FIRST | D_HASSLOT | (so << 16)
            else:
                info = D_HASSLOT | (so << 16)
                if last_ev:
                    info |= D_LAST
                    free_after.append(so)
            din = ie - ib
            nst = (din + rps - 1) // rps if din > 0 else 1
            for j in range(nst):
                cnt = max(min(rps, din - j * rps), 0)
                out.append((int(ib + j * rps), int(t), int(cnt | ((D_LASTSTEP | info) if j == nst - 1 else 0)), t in pending))
            free = sorted(free + free_after)
        steps[c] = out
        chunk_steps[c] = len(out)
    return {"chunk_node": chunk_node, "chunk_steps": chunk_steps, "steps": steps, "edge_meta": meta, "pending": pending, "far": far,
            "overflow": overflow}


class PartitionKernels:
    """The contracts of gnnome_hem_propose / gnnome_kway_gains (include/gnnome_hip.h) in plain Python: the checker backend of
    gnnome_amd/partition.py's host logic."""

    @staticmethod
    def hem_propose(ptr, adj, wgt, vwgt, match, max_vwgt):
        P, A, W, VW, M = (t.tolist() for t in (ptr, adj, wgt, vwgt, match))
        n = len(VW)
        out = [-1] * n
        for v in range(n):
            if M[v] >= 0:
                continue
            best, bw, bvw = -1, -1, 0
            for q in range(P[v], P[v + 1]):
                u = A[q]
                if u == v or M[u] >= 0 or VW[v] + VW[u] > max_vwgt:
                    continue
                if W[q] > bw or (W[q] == bw and (VW[u] < bvw or (VW[u] == bvw and u < best))):
                    best, bw, bvw = u, W[q], VW[u]
            out[v] = best
        return torch.tensor(out, dtype=torch.int32)

    @staticmethod
    def kway_gains(ptr, adj, wgt, label):
        P, A, W, L = (t.tolist() for t in (ptr, adj, wgt, label))
        n = len(L)
        bp, gn = [-1] * n, [0] * n
        for v in range(n):
            conn, internal = {}, 0
            for q in range(P[v], P[v + 1]):
                u = A[q]
                if u == v:
                    continue
                if L[u] == L[v]:
                    internal += W[q]
                elif L[u] in conn or len(conn) < 24:
                    conn[L[u]] = conn.get(L[u], 0) + W[q]
            best, bc = -1, -1
            for p_, c in conn.items():
                if c > bc or (c == bc and p_ < best):
                    best, bc = p_, c
            bp[v], gn[v] = best, (bc - internal if best >= 0 else 0)
        return torch.tensor(bp, dtype=torch.int32), torch.tensor(gn, dtype=torch.int32)
