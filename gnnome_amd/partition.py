"""Node clusters with a halo for mini-batch training (train.py:316-348).

The reference calls `dgl.metis_partition(g.long(), num_clusters, extra_cached_hops=1)` - METIS 5.1.0 inside DGL 0.8.1, two third-party
libraries that are not in this image - because a 10 GB GPU holds only ~2000-node clusters (hyperparameters.py:36-37).  On 288 GB of HBM
the full-graph step (train.py:319-330) always fits and is the regime this package is built for; for callers that keep the reference's
mini-batch regime this module provides the same call:

    parts = cluster_partition(graph, num_clusters, extra_cached_hops=1)     # dict: part id -> ClusterGraph
    in_deg, out_deg = features.stored_degrees(graph)                          # the FULL graph's degrees (ndata['in_deg'/'out_deg'])
    for sub in parts.values():
        x = features.partition_degree_features(in_deg, out_deg, sub.nid)       # get_partition_ne_features, train.py:125-135
        e = e_full[sub.eid];  y = y_full[sub.eid]
        logits = model(sub, x, e)                                           # get_bce_loss_partition, train.py:148-156

* `multilevel_partition` (round 5, the default): the published multilevel k-way scheme METIS implements (Karypis & Kumar 1998) on the
  device - coarsening by heavy-edge matching (csrc/partition.hip: proposals + handshake, contraction by sort + segmented sums), a
  greedy-growing partition of the coarsest graph (a few thousand vertices, on the host as in METIS), greedy k-way boundary refinement
  under the balance constraint (ufactor 1.03) while uncoarsening (csrc/partition.hip: gains; the moves of a pass go one way - to
  higher part ids in odd passes, to lower ones in even passes - so that no two neighbours swap).  Deterministic: a function of the graph.
  Not METIS itself (a randomised heuristic nobody can reproduce bit for bit): `oracle/metis_oracle.py` restates the scheme sequentially and
  `tests/test_partition.py` holds this implementation to its cut quality (and to the contiguous-range cut on layout-ordered graphs).
* `grow_regions` (rounds 2-4): multi-source breadth-first region growing with a size cap - connected, balanced, cut not minimised.
* the halo (`halo="dgl"`, the default): DGL's `partition_graph_with_halo` rule as its source states it - a part's own nodes first, then hop
  by hop the SOURCES of the in-edges of the nodes collected so far; the subgraph's edges are exactly those in-edges (one hop: every edge
  whose destination is an inner node).  `halo="both"`: the induced subgraph on the part plus every node within the hops in either
  direction (rounds 2-4).  DGL itself could not be run here: its rule is pinned through a hand-derived fixture (tests/golden/g13_partition_halo.json).
"""
import ctypes

import torch

from . import _lib, ops
from .features import MaskedGraph, _edge_list_on, induced_subgraph
from .ops import _on, _ptr, _stream


class ClusterGraph(MaskedGraph):
    """A MaskedGraph whose nodes carry `inner_node` (bool[N']): True for the cluster's own nodes, False for the halo
    (DGL's ndata['inner_node']); nid / eid are the reference's ndata['_ID'] / edata['_ID']."""

    def __init__(self, sub, inner_node):
        super().__init__(sub.src, sub.dst, sub.num_nodes(), sub.nid, sub.eid, sub.views)
        self.inner_node = inner_node


def grow_regions(src, dst, num_nodes, num_clusters, slack=1.05, generator=None):
    """label[N] in [0, num_clusters): seeds spread evenly over the node ids, regions grown one breadth-first level at a time
    over the undirected graph, a region stops taking nodes at ceil(N / K) * slack; nodes no region reached (other
    components, or walled in by full regions) seed further growth from the emptiest regions.  Deterministic."""
    dev = src.device
    n, k = int(num_nodes), int(num_clusters)
    cap = int(-(-n // k) * slack) + 1
    u = torch.cat([src, dst]).long()
    v = torch.cat([dst, src]).long()
    label = torch.full((n,), -1, dtype=torch.long, device=dev)
    seeds = (torch.arange(k, device=dev) * n // k + n // (2 * k)).clamp_(max=n - 1)
    label[seeds] = torch.arange(k, device=dev)
    size = torch.bincount(label[label >= 0], minlength=k)
    big = torch.iinfo(torch.long).max
    while True:
        lu, lv = label[u], label[v]
        open_edge = (lu >= 0) & (lv < 0)
        open_edge &= size[lu.clamp(min=0)] < cap
        if not bool(open_edge.any()):
            free = torch.nonzero(label < 0).squeeze(1)
            if free.numel() == 0:
                break
            # unreached nodes: hand the first of them to the emptiest region and keep growing from there
            r = int(torch.argmin(size))
            label[free[0]] = r
            size[r] += 1
            if int(size.min()) >= cap:      # every region is full: raise the cap rather than loop forever
                cap += max(1, cap // 20)
            continue
        cand = torch.full((n,), big, dtype=torch.long, device=dev)
        cand.scatter_reduce_(0, v[open_edge], lu[open_edge], "amin", include_self=True)   # smallest neighbouring region wins
        take = torch.nonzero(cand != big).squeeze(1)
        # respect the cap within the level: the first (cap - size) takers of every region, in node order
        lab = cand[take]
        order = torch.argsort(lab, stable=True)
        take, lab = take[order], lab[order]
        first = torch.searchsorted(lab, torch.arange(k, device=dev))
        rank = torch.arange(take.numel(), device=dev) - first[lab]
        ok = rank < (cap - size)[lab]
        label[take[ok]] = lab[ok]
        size = torch.bincount(label[label >= 0], minlength=k)
    return label


def undirected_csr(src, dst, num_nodes):
    """(ptr int32[n+1], adj int32[nnz], wgt int32[nnz]) on src's device: both directions, parallel / antiparallel edges merged
    (weight = multiplicity), self loops dropped, rows sorted by neighbour id."""
    n = int(num_nodes)
    a, b = src.long(), dst.long()
    keep = a != b
    a, b = a[keep], b[keep]
    keys, counts = torch.unique(torch.cat([a * n + b, b * n + a]), return_counts=True)
    row, col = keys // n, keys % n
    ptr = torch.zeros(n + 1, dtype=torch.int64, device=src.device)
    ptr[1:] = torch.cumsum(torch.bincount(row, minlength=n), 0)
    return ptr.int(), col.int().contiguous(), counts.int().contiguous()


class _HipKernels:
    """The two inner loops on the MI355X (csrc/partition.hip); tests inject a torch restatement of the same contracts (tests/cpu_ops.py) to run the
    host logic on the CPU, as engine.py / dist.py do with their `ops`."""

    @staticmethod
    def hem_propose(ptr, adj, wgt, vwgt, match, max_vwgt):
        lib = _lib.load()
        n, dev = ptr.numel() - 1, ptr.device
        prop = torch.empty(n, dtype=torch.int32, device=dev)
        with _on(dev):
            _lib.check(lib.gnnome_hem_propose(_ptr(ptr), _ptr(adj), _ptr(wgt), _ptr(vwgt), _ptr(match), n, int(max_vwgt), _ptr(prop), _stream(dev)),
                       "hem_propose")
        return prop

    @staticmethod
    def kway_gains(ptr, adj, wgt, label):
        lib = _lib.load()
        n, dev = label.numel(), label.device
        best_part = torch.empty(n, dtype=torch.int32, device=dev)
        gain = torch.empty(n, dtype=torch.int32, device=dev)
        with _on(dev):
            _lib.check(lib.gnnome_kway_gains(_ptr(ptr), _ptr(adj), _ptr(wgt), _ptr(label), n, _ptr(best_part), _ptr(gain), _stream(dev)), "kway_gains")
        return best_part, gain


def _greedy_growing_host(ptr, adj, wgt, vwgt, k):
    """_greedy_growing_py in the library (gnnome_greedy_growing_host: host memory in, host memory out, no launch)."""
    import ctypes
    lib = _lib.load()
    p, a, w, vw = (t.detach().to("cpu", torch.int32).contiguous() for t in (ptr, adj, wgt, vwgt))
    n = vw.numel()
    label = torch.empty(n, dtype=torch.int32)
    as_p = lambda t: ctypes.c_void_p(t.data_ptr() if t.numel() else None)  # noqa: E731
    _lib.check(lib.gnnome_greedy_growing_host(as_p(p), as_p(a), as_p(w), as_p(vw), n, int(k), as_p(label)), "greedy_growing_host")
    return label


_HipKernels.greedy_growing = staticmethod(_greedy_growing_host)


def _hem_level(ptr, adj, wgt, vwgt, max_vwgt, rounds=8, kernels=_HipKernels):
    """Heavy-edge matching by proposals and handshakes -> cmap int64[n] (coarse id of every vertex), nc."""
    n, dev = ptr.numel() - 1, ptr.device
    match = torch.full((n,), -1, dtype=torch.int32, device=dev)
    ids = torch.arange(n, device=dev, dtype=torch.int32)
    for _ in range(rounds):
        prop = kernels.hem_propose(ptr, adj, wgt, vwgt, match, max_vwgt)
        has = prop >= 0
        if not bool(has.any()):
            break
        mutual = has & (prop[prop.clamp(min=0).long()] == ids)
        match = torch.where(mutual, prop, match)
    match = torch.where(match < 0, ids, match)                       # the rest stay single
    rep = torch.minimum(ids, match).long()                           # a pair is named by its smaller id
    uniq, cmap = torch.unique(rep, return_inverse=True)              # coarse ids in the order of the representatives
    return cmap, int(uniq.numel())


def _contract(ptr, adj, wgt, vwgt, cmap, nc):
    n, dev = ptr.numel() - 1, ptr.device
    row = torch.repeat_interleave(torch.arange(n, device=dev), (ptr[1:] - ptr[:-1]).long())
    cu, cv = cmap[row], cmap[adj.long()]
    keep = cu != cv
    keys, inv = torch.unique(cu[keep] * nc + cv[keep], return_inverse=True)
    w = torch.zeros(keys.numel(), dtype=torch.int64, device=dev).scatter_add_(0, inv, wgt[keep].long())
    crow = keys // nc
    cptr = torch.zeros(nc + 1, dtype=torch.int64, device=dev)
    cptr[1:] = torch.cumsum(torch.bincount(crow, minlength=nc), 0)
    cvw = torch.zeros(nc, dtype=torch.int64, device=dev).scatter_add_(0, cmap, vwgt.long())
    return cptr.int(), (keys % nc).int().contiguous(), w.int().contiguous(), cvw.int()


def _greedy_growing_py(ptr, adj, wgt, vwgt, k):
    """(The statement of gnnome_greedy_growing_host - csrc/partition.hip runs it in C++, same labels bit for bit; the checker backend of the CPU
    tests runs this one.)  The coarsest graph (a few thousand vertices) on the host, as METIS does it: k - 1 regions grown one after the other from the free
    vertex with the smallest id, always taking the free vertex most heavily connected to the region (ties: the one that became a neighbour of
    the region first - breadth-first, compact regions; by smaller id a grid is cut into strips), until the region holds its share of the
    weight; the rest is the last part."""
    import heapq
    ptr, adj, wgt, vwgt = (t.cpu().tolist() for t in (ptr, adj, wgt, vwgt))
    n = len(vwgt)
    label, assigned_w, total = [-1] * n, 0, sum(vwgt)
    next_free = 0
    for p in range(k - 1):
        target = (total - assigned_w) / (k - p)
        heap, conn, size, tick = [], {}, 0, 0
        first_seen = {}
        while size < target:
            while heap and (label[heap[0][2]] >= 0 or -heap[0][0] != conn[heap[0][2]]):
                heapq.heappop(heap)
            if heap:
                v = heapq.heappop(heap)[2]
            else:
                while next_free < n and label[next_free] >= 0:
                    next_free += 1
                if next_free == n:
                    break
                v = next_free
            label[v] = p
            size += vwgt[v]
            for q in range(ptr[v], ptr[v + 1]):
                u = adj[q]
                if label[u] < 0:
                    conn[u] = conn.get(u, 0) + wgt[q]
                    if u not in first_seen:
                        first_seen[u] = tick
                        tick += 1
                    heapq.heappush(heap, (-conn[u], first_seen[u], u))
        assigned_w += size
    return torch.tensor([k - 1 if x < 0 else x for x in label], dtype=torch.int32)


def _cut_weight(ptr, adj, wgt, label):
    row_len = (ptr[1:] - ptr[:-1]).long()
    row = torch.repeat_interleave(torch.arange(label.numel(), device=label.device), row_len)
    return int(wgt[label[row] != label[adj.long()]].long().sum()) // 2


def _refine(ptr, adj, wgt, vwgt, label, k, max_pw, passes=12, kernels=_HipKernels):
    """Greedy k-way boundary refinement (Karypis & Kumar, JPDC 1998, section 4) as parallel passes: gains from the kernel, then per target part the
    candidates in order of decreasing gain as far as the part's weight allows; a pass moves vertices only towards higher (odd passes) or
    lower (even passes) part ids, so two neighbours never swap; what is returned is the best labelling seen - feasible (no part above max_pw) before smaller cut."""
    n, dev = label.numel(), label.device
    vw = vwgt.long()
    def excess(lab):   # weight above the limit in the heaviest part: 0 = the labelling is feasible (ufactor holds)
        return max(0, int(torch.zeros(k, dtype=torch.int64, device=dev).scatter_add_(0, lab.long(), vw).max()) - max_pw)

    # best = feasibility first, cut second (the smallest cut alone kept a greedy-growing overshoot that the balancing moves - negative
    # gains out of an overweight part - had already repaired)
    best_label, best_key = label.clone(), (excess(label), _cut_weight(ptr, adj, wgt, label))
    stale = 0
    for it in range(passes):
        best_part, gain = kernels.kway_gains(ptr, adj, wgt, label)
        pw = torch.zeros(k, dtype=torch.int64, device=dev).scatter_add_(0, label.long(), vw)
        up = (it % 2) == 0
        cand = (best_part >= 0) & ((best_part > label) if up else (best_part < label))
        # positive gain, or zero gain towards a lighter part (helps the next passes across plateaus), or any gain out of an overweight part
        tgt = best_part.clamp(min=0).long()
        cand &= (gain > 0) | ((gain == 0) & (pw[tgt] + vw < pw[label.long()])) | (pw[label.long()] > max_pw)
        idx = torch.nonzero(cand).squeeze(1)
        if idx.numel() == 0:
            if stale:
                break
            stale = 1
            continue
        t, g_ = tgt[idx], gain[idx].long()
        order = torch.argsort(t * (1 << 32) + ((1 << 31) - 1 - g_.clamp(min=-(1 << 30))), stable=True)   # by target, then by decreasing gain, then by id
        idx, t = idx[order], t[order]
        w_sorted = vw[idx]
        csum = torch.cumsum(w_sorted, 0)
        first = torch.searchsorted(t, torch.arange(k, device=dev))
        base = torch.cat([torch.zeros(1, dtype=csum.dtype, device=dev), csum])[first]   # weight of the candidates sorted before part t's first one
        within = csum - base[t]                                         # weight moved into t up to and including this candidate
        ok = within <= (max_pw - pw)[t]
        moved = idx[ok]
        if moved.numel() == 0:
            if stale:
                break
            stale = 1
            continue
        label = label.clone()
        label[moved] = best_part[moved]
        key = (excess(label), _cut_weight(ptr, adj, wgt, label))
        if key < best_key:
            best_key, best_label, stale = key, label.clone(), 0
        else:
            stale += 1
            if stale >= 3:
                break
    return best_label


def multilevel_partition(src, dst, num_nodes, num_clusters, ufactor=1.03, kernels=_HipKernels):
    """label int64[N] in [0, num_clusters): the multilevel k-way scheme (module docstring) on the device of `src`."""
    n, k, dev = int(num_nodes), int(num_clusters), src.device
    if k <= 1 or n == 0:
        return torch.zeros(n, dtype=torch.long, device=dev)
    ptr, adj, wgt = undirected_csr(src, dst, n)
    vwgt = torch.ones(n, dtype=torch.int32, device=dev)
    levels = []
    coarsen_to = max(30 * k, 200)
    while ptr.numel() - 1 > coarsen_to:
        nv = ptr.numel() - 1
        max_vwgt = max(1, int(1.5 * n / coarsen_to))
        cmap, nc = _hem_level(ptr, adj, wgt, vwgt, max_vwgt, kernels=kernels)
        if nc > 0.95 * nv:          # the matching no longer shrinks the graph (stars, isolated vertices)
            break
        levels.append((ptr, adj, wgt, vwgt, cmap))
        ptr, adj, wgt, vwgt = _contract(ptr, adj, wgt, vwgt, cmap, nc)
    max_pw = int(ufactor * n / k) + 1
    label = getattr(kernels, "greedy_growing", _greedy_growing_py)(ptr, adj, wgt, vwgt, k).to(dev)
    label = _refine(ptr, adj, wgt, vwgt, label, k, max_pw, kernels=kernels)
    for fptr, fadj, fwgt, fvw, cmap in reversed(levels):
        label = label[cmap].contiguous()
        label = _refine(fptr, fadj, fwgt, fvw, label, k, max_pw, kernels=kernels)
    return label.long()


def edge_cut(src, dst, label):
    return int((label[src.long()] != label[dst.long()]).sum())


def _dgl_halo(src, dst, n, inner, hops):
    """DGL's partition_graph_with_halo for ONE part (module docstring): -> (nid: inner nodes ascending, then the halo nodes hop by hop, each
    hop ascending; eid: the in-edges of the nodes collected hop by hop, ascending edge id within a hop)."""
    s_l, d_l = src.long(), dst.long()
    seen = inner.clone()
    nid = [torch.nonzero(inner).squeeze(1)]
    eids, frontier = [], inner
    if int(hops) == 0:
        return nid[0], torch.nonzero(inner[d_l] & inner[s_l]).squeeze(1)
    for _ in range(int(hops)):
        e_in = torch.nonzero(frontier[d_l]).squeeze(1)                  # the in-edges of the frontier
        eids.append(e_in)
        new = torch.zeros_like(seen)
        new[s_l[e_in]] = True
        new &= ~seen
        nid.append(torch.nonzero(new).squeeze(1))
        seen |= new
        frontier = new
    return torch.cat(nid), torch.cat(eids)


def _dgl_one_hop_parts(src, dst, n, label, k, device):
    """_dgl_halo(..., hops=1) for every part at once: a part's edges are the in-edges of its inner nodes (ascending edge id), its nodes the inner
    ones (ascending) followed by the outside sources of those edges (ascending).  Sorts over the whole graph, then slices."""
    s_l, d_l, lab = src.long(), dst.long(), label.long()
    part_e = lab[d_l]                                            # the part every edge belongs to
    e_order = torch.argsort(part_e, stable=True)                 # edges by part, ascending id inside
    e_ptr = torch.zeros(k + 1, dtype=torch.long, device=device)
    e_ptr[1:] = torch.cumsum(torch.bincount(part_e, minlength=k), 0)
    n_order = torch.argsort(lab, stable=True)                    # nodes by part, ascending id inside
    n_ptr = torch.zeros(k + 1, dtype=torch.long, device=device)
    n_ptr[1:] = torch.cumsum(torch.bincount(lab, minlength=k), 0)
    rank = torch.empty(n, dtype=torch.long, device=device)       # a node's position among the inner nodes of its part
    rank[n_order] = torch.arange(n, device=device) - n_ptr[lab[n_order]]
    outside = lab[s_l] != part_e
    keys = torch.unique(part_e[outside] * n + s_l[outside])      # (part, outside source), ascending
    h_part, h_node = keys // n, keys % n
    h_ptr = torch.zeros(k + 1, dtype=torch.long, device=device)
    h_ptr[1:] = torch.cumsum(torch.bincount(h_part, minlength=k), 0)
    n_inner = n_ptr[1:] - n_ptr[:-1]
    pos = torch.searchsorted(keys, part_e * n + s_l)             # (meaningful where `outside`)
    loc_src = torch.where(outside, n_inner[part_e] + pos - h_ptr[part_e], rank[s_l]).int()
    loc_dst = rank[d_l].int()
    e_lo, n_lo, h_lo = e_ptr.tolist(), n_ptr.tolist(), h_ptr.tolist()
    parts = {}
    for p in range(k):
        ni = n_lo[p + 1] - n_lo[p]
        if ni == 0:
            continue
        eid = e_order[e_lo[p]:e_lo[p + 1]]
        nid = torch.cat([n_order[n_lo[p]:n_lo[p + 1]], h_node[h_lo[p]:h_lo[p + 1]]])
        sub_src, sub_dst = loc_src[eid].contiguous(), loc_dst[eid].contiguous()
        views = ops.GraphViews(sub_src, sub_dst, int(nid.numel()), validate=False)
        sub = MaskedGraph(sub_src, sub_dst, int(nid.numel()), nid, eid, views)
        inner = torch.zeros(int(nid.numel()), dtype=torch.bool, device=device)
        inner[:ni] = True
        parts[p] = ClusterGraph(sub, inner)
    return parts


def cluster_partition(graph, num_clusters, extra_cached_hops=1, device=None, method="multilevel", halo="dgl", one_pass=True):
    """-> {part id: ClusterGraph}; see the module docstring.  method: "multilevel" | "region"; halo: "dgl" | "both".  one_pass: the reference's
    own call (one hop, DGL's rule: train.py:334) cuts ALL parts out of the graph with a handful of sorts (_dgl_one_hop_parts) instead of one scan
    of the whole graph per part (k = N / 2000 parts: hyperparameters.py:36) - same sub-graphs, same order of nodes and edges."""
    device = device or (graph.device if isinstance(graph, ops.GraphViews) else torch.device("cuda", torch.cuda.current_device()))
    src, dst, n = _edge_list_on(graph, device)
    if method == "multilevel":
        label = multilevel_partition(src, dst, n, num_clusters)
    elif method == "region":
        label = grow_regions(src, dst, n, num_clusters)
    else:
        raise ValueError(f"method={method!r} not in ('multilevel', 'region')")
    if halo not in ("dgl", "both"):
        raise ValueError(f"halo={halo!r} not in ('dgl', 'both')")
    parts = {}
    s_l, d_l = src.long(), dst.long()
    if halo == "dgl" and int(extra_cached_hops) == 1 and one_pass:
        return _dgl_one_hop_parts(src, dst, n, label, int(num_clusters), device)
    for p in range(int(num_clusters)):
        inner = label == p
        if not bool(inner.any()):
            continue
        if halo == "dgl":
            nid, eid = _dgl_halo(src, dst, n, inner, extra_cached_hops)
            new_id = torch.full((n,), -1, dtype=torch.long, device=device)
            new_id[nid] = torch.arange(nid.numel(), device=device)
            sub_src, sub_dst = new_id[s_l[eid]].int(), new_id[d_l[eid]].int()
            views = ops.GraphViews(sub_src, sub_dst, int(nid.numel()), validate=False)
            sub = MaskedGraph(sub_src, sub_dst, int(nid.numel()), nid, eid, views)
            parts[p] = ClusterGraph(sub, inner[nid])
            continue
        keep = inner.clone()
        for _ in range(int(extra_cached_hops)):
            grow = torch.zeros_like(keep)
            grow[d_l[keep[s_l]]] = True      # successors of kept nodes
            grow[s_l[keep[d_l]]] = True      # predecessors of kept nodes
            keep |= grow
        sub = induced_subgraph((src, dst, n), keep, device)
        parts[p] = ClusterGraph(sub, inner[sub.nid])
    return parts
