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
    return ops.degree_features(views, reverse)


def edge_features(overlap_length, overlap_similarity):
    """e[E,2] = [zscore(overlap_length) | overlap_similarity] (utils/data_utils.py:31-41), on the device of its inputs."""
    return ops.edge_features(overlap_length, overlap_similarity)


degree_features_hip, edge_features_hip = degree_features, edge_features   # earlier names
