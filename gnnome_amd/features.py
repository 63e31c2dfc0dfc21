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
    """x[N,2] = [zscore(in_degree) | zscore(out_degree)] of the graph the views describe (columns swapped for
    `reverse`, i.e. for dgl.reverse(g) - train.py:116-117), on the device.  `graph` = anything
    gnnome_amd.graph.views_for accepts: DGLGraph, (src, dst, N), or GraphViews."""
    views = graph if isinstance(graph, ops.GraphViews) else views_for(graph, device or torch.device("cuda", torch.cuda.current_device()))
    return ops.degree_features(views, reverse)


def edge_features(overlap_length, overlap_similarity):
    """e[E,2] = [zscore(overlap_length) | overlap_similarity] (utils/data_utils.py:31-41), on the device of its inputs."""
    return ops.edge_features(overlap_length, overlap_similarity)


degree_features_hip, edge_features_hip = degree_features, edge_features   # earlier names
