"""Caller-side feature preparation on the device: what the reference drivers compute between loading a graph and
calling the model (inference.py:413-420, train.py:112-122, utils/data_utils.py:31-41).

Degrees come straight off the CSR pointers of the graph views that the model call needs anyway
(gnnome_degree_features_f32), the edge features from the parser's overlap lengths and similarities
(gnnome_edge_features_f32): a harness goes from edge list to logits without touching the host.  There is no host
version in this package; the oracle (oracle/symgated_oracle.py) holds the torch restatement the tests check against."""
import torch

from . import ops
from .graph import views_for


def degree_features(graph, reverse=False, device=None):
    """get_full_ne_features(g, reverse)[0] (train.py:112-122; inference.py:416-420): x[N,2] = [zscore(in_deg) |
    zscore(out_deg)], columns swapped when `reverse`, on the device.

    `reverse` means what it means in the reference.  There the degrees are STORED node features (ndata['in_deg'],
    ndata['out_deg'], written by the graph parser) that dgl.reverse(g, True, True) copies without recomputing, so the
    symmetry loss's second pass - `g = dgl.reverse(g, True, True); get_full_ne_features(g, reverse=True)`
    (train.py:165-166) - returns [out | in] of the ORIGINAL graph.  Accordingly:
      * a graph object that carries ndata['in_deg'] / ndata['out_deg'] (a GNNome DGLGraph, reversed or not): those
        stored values are z-scored, exactly like the reference;
      * GraphViews / (src, dst, N) / any other graph: degrees are counted from the edge list the views were built from;
        `GraphViews.reversed()` shares that edge list, so `degree_features(views.reversed(), reverse=True)` is the
        reference's second pass.  (A (dst, src, N) tuple - an edge list that was itself reversed by hand - has its own
        degrees; pass reverse=False for it.)"""
    device = device or (graph.device if isinstance(graph, ops.GraphViews) else torch.device("cuda", torch.cuda.current_device()))
    nd = getattr(graph, "ndata", None)
    if nd is not None and "in_deg" in nd and "out_deg" in nd:
        cols = []
        for key in (("out_deg", "in_deg") if reverse else ("in_deg", "out_deg")):
            d = torch.as_tensor(nd[key]).to(device=device, dtype=torch.float32)
            cols.append(((d - d.mean()) / d.std()).unsqueeze(1))   # torch.std: unbiased, as in the reference
        return torch.cat(cols, 1).contiguous()
    views = graph if isinstance(graph, ops.GraphViews) else views_for(graph, device)
    x = ops.degree_features(views, reverse)
    views.check_range()   # a fresh graph's deferred endpoint check (GraphViews validate="lazy"): clamped endpoints never pass silently
    return x


def stored_degrees(graph, device=None):
    """(in_deg, out_deg) float32[N] of `graph` as the reference's parser stores them in ndata (graph_parser.py: the degrees
    of the FULL graph, which sub-graphs index into), counted on the device from the views' CSR pointers."""
    device = device or (graph.device if isinstance(graph, ops.GraphViews) else torch.device("cuda", torch.cuda.current_device()))
    views = graph if isinstance(graph, ops.GraphViews) else views_for(graph, device)
    views.check_range()
    ind = (views.in_ptr[1:] - views.in_ptr[:-1]).float()
    outd = (views.out_ptr[1:] - views.out_ptr[:-1]).float()
    return (outd, ind) if views.transposed else (ind, outd)


def partition_degree_features(full_in_deg, full_out_deg, nid, reverse=False):
    """get_partition_ne_features(sub_g, g, reverse)[0] (train.py:125-135): the FULL graph's stored degrees of the sub-graph's
    nodes (`nid` = sub_g.ndata['_ID']: MaskedGraph.nid, Cluster.nid), z-scored with the mean and unbiased std of THOSE nodes -
    neither a slice of the full graph's z-scored table (different mean / std) nor the degrees recounted on the induced
    sub-graph (different counts).  The strand-wise mask (train.py:311-313) and the METIS mini-batches (:339-346) both feed
    the model this."""
    nid = torch.as_tensor(nid).long()
    cols = []
    for d in ((full_out_deg, full_in_deg) if reverse else (full_in_deg, full_out_deg)):
        d = torch.as_tensor(d).to(dtype=torch.float32)
        d = d[nid.to(d.device)]
        cols.append(((d - d.mean()) / d.std()).unsqueeze(1))
    return torch.cat(cols, 1).contiguous()


def edge_features(overlap_length, overlap_similarity):
    """e[E,2] = [zscore(overlap_length) | overlap_similarity] (utils/data_utils.py:31-41), on the device of its inputs."""
    return ops.edge_features(overlap_length, overlap_similarity)


class MaskedGraph:
    """Result of mask_graph_strandwise: the induced subgraph on the kept reads (both strands of each), relabelled.

    src, dst     int32[E']  endpoints in the subgraph's own numbering (kept nodes in ascending original order)
    nid          int64[N']  original node id of subgraph node i   (the reference's sub_g.ndata[dgl.NID])
    eid          int64[E']  original edge id of subgraph edge k   (sub_g.edata[dgl.EID]; original order kept)
    views        GraphViews of the subgraph, ready for model(views, x[nid], e[eid])
    Works as the `graph` argument of the model (edges() / num_nodes())."""

    def __init__(self, src, dst, num_nodes, nid, eid, views):
        self.src, self.dst, self._n, self.nid, self.eid, self.views = src, dst, num_nodes, nid, eid, views

    def edges(self):
        return self.src, self.dst

    def num_nodes(self):
        return self._n

    def num_edges(self):
        return int(self.src.numel())


def _edge_list_on(graph, device):
    if isinstance(graph, ops.GraphViews):
        if graph.transposed:
            raise ValueError("take the subgraph of the original orientation and reverse the result")
        src = torch.empty(graph.num_edges, dtype=torch.int32, device=device)
        dst = torch.empty_like(src)
        src[graph.srt_eid.long()], dst[graph.srt_eid.long()] = graph.srt_src, graph.srt_dst     # back to edge-id order
        return src, dst, graph.num_nodes
    from .graph import edge_list
    src, dst, n = edge_list(graph)
    return src.to(device=device, dtype=torch.int32), dst.to(device=device, dtype=torch.int32), n


def induced_subgraph(graph, keep, device=None):
    """dgl.node_subgraph(g, keep, store_ids=True) on the device: `keep` bool[N]; kept nodes renumbered in ascending order,
    the edges whose two endpoints are kept in their original order, original ids stored (MaskedGraph.nid / .eid)."""
    device = device or (graph.device if isinstance(graph, ops.GraphViews) else torch.device("cuda", torch.cuda.current_device()))
    src, dst, n = _edge_list_on(graph, device)
    keep = keep.to(device)
    new_id = torch.cumsum(keep, 0, dtype=torch.int32) - 1
    eid = torch.nonzero(keep[src.long()] & keep[dst.long()]).squeeze(1)
    nid = torch.nonzero(keep).squeeze(1)
    s2, d2 = new_id[src[eid].long()].contiguous(), new_id[dst[eid].long()].contiguous()
    views = ops.GraphViews(s2, d2, int(nid.numel()), validate=False)
    return MaskedGraph(s2, d2, int(nid.numel()), nid, eid, views)


def mask_graph_strandwise(graph, fraction, device=None, keep_half=None):
    """train.py:91-100 on the device: keep each READ with probability `fraction` - both its strands, nodes 2r and 2r+1 -
    and return the induced subgraph (dgl.node_subgraph(g, keep, store_ids=True): kept nodes renumbered in ascending order,
    the edges whose two endpoints are kept in their original order, original ids stored).  `graph`: (src, dst, N), a
    DGLGraph, or GraphViews.  `keep_half` (bool[N/2]) overrides the random draw (the reference draws
    torch.rand(N // 2, device=device) < fraction; pass that tensor to reproduce its stream)."""
    device = device or (graph.device if isinstance(graph, ops.GraphViews) else torch.device("cuda", torch.cuda.current_device()))
    n = graph.num_nodes if isinstance(graph, ops.GraphViews) else (graph[2] if isinstance(graph, (tuple, list)) else graph.num_nodes())
    n = int(n)
    if n % 2:
        raise ValueError("nodes come in (read, reverse complement) pairs: N must be even")
    if keep_half is None:
        keep_half = torch.rand(n // 2, device=device) < fraction
    return induced_subgraph(graph, keep_half.to(device).repeat_interleave(2), device)


degree_features_hip, edge_features_hip = degree_features, edge_features   # earlier names
