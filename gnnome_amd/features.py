"""Caller-side feature preparation, mirrored so the repo's own harnesses (bench.py, smoke) can feed the
model exactly what the reference drivers feed it.  torch only - this is plumbing around the path."""
import torch


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
