"""Locality-restoring node order for graphs whose node ids do not follow the layout.

The reference numbers nodes in S-line order of the GFA (graph_parser.py:174-181: read r -> nodes 2r and 2r + 1); nothing makes
that the order of the reads along the genome.  This package's multi-GPU partition is by node RANGE (gnnome_amd/dist.py) and its
gathers live on L2 reuse, so shuffled ids cost: 87.5 % of the edges cut at 8 ranks instead of 1 %, the forward 5.5 instead of
4.7 ms on one GPU.  `locality_order` renumbers the READS (both strands of a read stay adjacent: 2r', 2r' + 1) once per graph:

  1. the undirected read graph as a sorted, de-duplicated CSR (torch: one unique over 2E keys);
  2. keep the entries whose endpoints have a common neighbour (gnnome_adjacency_support): overlaps are locally transitive,
     repeat-induced long-range edges are not - 1 % of those is enough to ruin any breadth-first order taken over all edges;
  3. breadth-first levels over the kept entries, twice (gnnome_bfs_levels): from the smallest id of every component, then from
     the far node that pass found - a start at one END of a contig gives levels that sweep it once;
  4. reads without a kept edge take the smallest key among their neighbours (any edge), or go last;
  5. new read id = rank of (level key, old id) - a function of the graph alone (levels do not depend on thread timing).

Nothing a caller of the reference sees changes: `x` goes in and logits come out in the caller's numbering
(GraphViews.node_perm / node_gather; engine.run_stack gathers x rows through the encoder's gather argument).
"""
import torch

from . import _lib
from .ops import _on, _ptr, _stream


def read_adjacency(src, dst, num_reads, pair=True):
    """(ptr int32[R+1], adj int32[nnz], row int32[nnz]): undirected, no self loops, rows sorted and de-duplicated."""
    a = (src.long() >> 1) if pair else src.long()
    b = (dst.long() >> 1) if pair else dst.long()
    keep = a != b
    a, b = a[keep], b[keep]
    keys = torch.unique(torch.cat([a * num_reads + b, b * num_reads + a]))
    row, col = (keys // num_reads), (keys % num_reads)
    ptr = torch.zeros(num_reads + 1, dtype=torch.int64, device=src.device)
    ptr[1:] = torch.cumsum(torch.bincount(row, minlength=num_reads), 0)
    return ptr.int(), col.int().contiguous(), row.int().contiguous()


def adjacency_support(ptr, adj, row):
    """uint8[nnz]: 1 where the entry's endpoints share a neighbour."""
    out = torch.empty(adj.numel(), dtype=torch.uint8, device=adj.device)
    lib = _lib.load()
    with _on(adj.device):
        _lib.check(lib.gnnome_adjacency_support(_ptr(ptr), _ptr(adj), _ptr(row), ptr.numel() - 1, adj.numel(), _ptr(out), _stream(adj.device)),
                   "adjacency_support")
    return out


def bfs_levels(ptr, adj, seeds=None, num_seeds=None):
    """-> (level_key int32[R] (-1: no neighbours), far_node int32[R], num_components int32[1]) of the CSR graph (ptr, adj)."""
    R, dev = ptr.numel() - 1, ptr.device
    key = torch.empty(R, dtype=torch.int32, device=dev)
    far = torch.zeros(max(R, 1), dtype=torch.int32, device=dev)
    ncomp = torch.zeros(1, dtype=torch.int32, device=dev)
    ws = torch.empty(2 * max(R, 1), dtype=torch.int32, device=dev)
    lib = _lib.load()
    with _on(dev):
        _lib.check(lib.gnnome_bfs_levels(_ptr(ptr), _ptr(adj), R, _ptr(seeds), _ptr(num_seeds), _ptr(key), _ptr(ws), _ptr(far), _ptr(ncomp),
                                         _stream(dev)), "bfs_levels")
    ws.record_stream(torch.cuda.current_stream(dev))
    return key, far, ncomp


def locality_order(src, dst, num_nodes, pair=None, return_stats=False):
    """int64[num_nodes] on src's device: perm[old node id] = new node id.  pair: reads are node pairs (2r, 2r + 1) that must stay
    adjacent (default: whenever num_nodes is even - the reference's graphs always are)."""
    dev = src.device
    if pair is None:
        pair = num_nodes % 2 == 0
    R = num_nodes // 2 if pair else num_nodes
    if R == 0 or src.numel() == 0:
        return torch.arange(num_nodes, device=dev)
    ptr, adj, row = read_adjacency(src, dst, R, pair)
    sup = adjacency_support(ptr, adj, row).bool()
    kept_row = row[sup].long()
    kptr = torch.zeros(R + 1, dtype=torch.int64, device=dev)
    kptr[1:] = torch.cumsum(torch.bincount(kept_row, minlength=R), 0)
    kptr, kadj = kptr.int(), adj[sup].contiguous()
    _, far, ncomp = bfs_levels(kptr, kadj)                       # pass 1: finds a far end of every component
    key, _, _ = bfs_levels(kptr, kadj, seeds=far, num_seeds=ncomp)   # pass 2: levels from those ends
    key = key.long()
    # reads outside the kept graph: next to their best-placed neighbour (over all edges), else at the end
    lone = key < 0
    big = R + 1   # above every level key; (big * R + id stays far inside int64 - int64.max // 4 here wrapped around for any R > 4)
    if bool(lone.any()):
        nb_key = torch.where(key[adj.long()] < 0, torch.full_like(adj, big, dtype=torch.int64), key[adj.long()])
        best = torch.full((R,), big, dtype=torch.int64, device=dev).scatter_reduce(0, row.long(), nb_key, reduce="amin", include_self=True)
        key = torch.where(lone, best, key)
    order = torch.argsort(key * R + torch.arange(R, device=dev), stable=True)       # (level key, old id): unique keys
    new_read = torch.empty(R, dtype=torch.int64, device=dev)
    new_read[order] = torch.arange(R, device=dev)
    if pair:
        old = torch.arange(num_nodes, device=dev)
        perm = 2 * new_read[old >> 1] + (old & 1)
    else:
        perm = new_read
    if return_stats:
        return perm, {"reads": R, "adjacency_entries": int(adj.numel()), "supported_entries": int(sup.sum()), "components": int(ncomp),
                      "levels": int(key[~lone].max()) + 1 if bool((~lone).any()) else 0, "lone_reads": int(lone.sum())}
    return perm


def auto_order(src, dst, num_nodes, span_fraction=1 / 16):
    """node_order="auto": (perm or None, info).  One device statistic decides - the mean |src - dst| of the edge list; beyond
    span_fraction * N the ids do not follow the layout (a layout-ordered assembly graph: ~2 x its degree; a shuffled one or a uniform
    random one: ~N / 3) and locality_order runs; its result is kept only if it at least halved that span (a graph without locality,
    e.g. a uniform random one, is left alone).  info: span_before, span_after, order_ms, decision."""
    import time
    before = mean_edge_span(src, dst)
    info = {"span_before": before, "threshold": span_fraction * num_nodes, "decision": "input"}
    if before <= span_fraction * num_nodes:
        return None, info
    torch.cuda.synchronize(src.device) if src.is_cuda else None
    t0 = time.perf_counter()
    perm = locality_order(src, dst, num_nodes)
    after = mean_edge_span(perm[src.long()], perm[dst.long()])
    info["order_ms"], info["span_after"] = (time.perf_counter() - t0) * 1e3, after
    if after > 0.5 * before:
        return None, info
    info["decision"] = "locality"
    return perm, info


def mean_edge_span(src, dst):
    """mean |src - dst| of an edge list: the locality figure `views_for(..., node_order="auto")` looks at."""
    return float((src.long() - dst.long()).abs().float().mean()) if src.numel() else 0.0
