"""Caller-side feature preparation: what the reference drivers compute between loading a graph and calling the
model (inference.py:413-420, train.py:112-122, utils/data_utils.py:31-41).

`*_hip` are the device versions (gnnome_degree_features_f32 / gnnome_edge_features_f32): degrees come straight off
the CSR pointers of the graph views that the model call needs anyway, so a harness goes from edge list to logits
without touching the host.  The plain functions are the same arithmetic in torch for harnesses that prepare their
inputs on the host (bench.py's multi-rank setup, the tests' fixtures)."""
import torch

from . import ops
from .graph import views_for


def degree_features_hip(graph, reverse=False, device=None):
    """x[N,2] on the device; `graph` = anything gnnome_amd.graph.views_for accepts (DGLGraph, (src, dst, N), views)."""
    views = graph if isinstance(graph, ops.GraphViews) else views_for(graph, device or torch.device("cuda", torch.cuda.current_device()))
    return ops.degree_features(views, reverse)


def edge_features_hip(overlap_length, overlap_similarity):
    """e[E,2] on the device of its inputs."""
    return ops.edge_features(overlap_length, overlap_similarity)


def degree_features(src, dst, num_nodes, reverse=False):
    """x[N,2] = [zscore(in_degree) | zscore(out_degree)], torch.std's unbiased estimator
    (inference.py:416-420; train.py:112-122 swaps the columns for the reversed graph)."""
    src, dst = torch.as_tensor(src).long(), torch.as_tensor(dst).long()
    ind = torch.bincount(dst, minlength=num_nodes).float().unsqueeze(1)
    outd = torch.bincount(src, minlength=num_nodes).float().unsqueeze(1)
    ind = (ind - ind.mean()) / ind.std()
    outd = (outd - outd.mean()) / outd.std()
    return torch.cat((outd, ind), 1) if reverse else torch.cat((ind, outd), 1)


def edge_features(overlap_length, overlap_similarity):
    """e[E,2] = [zscore(overlap_length) | overlap_similarity] (utils/data_utils.py:33-38)."""
    ol = overlap_length.float()
    ol = (ol - ol.mean()) / ol.std()
    return torch.cat((ol.unsqueeze(-1), overlap_similarity.float().unsqueeze(-1)), dim=1)
