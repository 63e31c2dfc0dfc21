"""Synthetic assembly graphs shaped like GNNome's.

Structural conventions kept from the reference's graph builder:
  * N is even; node 2r is read r, node 2r+1 its reverse complement (graph_parser.py:174-181)
  * every edge (u, v) has a mate (v^1, u^1) carrying the same features (graph_parser.py:300-340)
  * e = [zscore(overlap_length), overlap_similarity] (utils/data_utils.py:31-41)
Edge ids are randomly permuted so the CSR build and the un-permute of the logits are exercised.
kind: "banded" (reads in layout order, 1 % long-range edges), "uniform" (no locality at all), "permuted" (the banded graph with
shuffled read ids: locality exists but the numbering hides it - gnnome_amd/node_order.py finds it again).
"""
import numpy as np
import torch


def _pairs_banded(reads, n_pairs, mean_deg, rng, rewire=0.01):
    d = np.minimum(rng.poisson(mean_deg, size=reads), int(4 * mean_deg)).astype(np.int64)
    # trim / pad the degree sequence to exactly n_pairs primary edges
    diff = int(d.sum()) - n_pairs
    while diff != 0:
        if diff > 0:
            idx = np.flatnonzero(d > 0)
            take = rng.choice(idx, size=min(diff, idx.size), replace=False)
            d[take] -= 1
        else:
            take = rng.choice(reads, size=min(-diff, reads), replace=False)
            d[take] += 1
        diff = int(d.sum()) - n_pairs
    r = np.repeat(np.arange(reads, dtype=np.int64), d)
    first = np.cumsum(d) - d
    off = np.arange(n_pairs, dtype=np.int64) - np.repeat(first, d) + 1
    t = (r + off) % reads
    rew = rng.random(n_pairs) < rewire
    t[rew] = rng.integers(0, reads, size=int(rew.sum()))
    strand = rng.integers(0, 2, size=n_pairs)
    return 2 * r + strand, 2 * t + strand


def make_graph(num_nodes, num_edges, seed=1, kind="banded"):
    """Returns dict(src int32[E], dst int32[E], num_nodes, e float32[E,2], y float32[E], pos_weight)."""
    assert num_nodes % 2 == 0 and num_edges % 2 == 0
    rng = np.random.default_rng(seed)
    half = num_edges // 2
    if kind in ("banded", "permuted"):
        u, v = _pairs_banded(num_nodes // 2, half, num_edges / num_nodes, rng)
        if kind == "permuted":
            # the banded graph with its READ ids shuffled (strands stay paired): what a GFA whose S lines are not in layout
            # order gives (graph_parser.py:174-181 numbers reads in S-line order).  A generator of its own, so that the
            # "banded" stream - and with it every other tensor of the graph - is the same as for kind="banded".
            rp = np.random.default_rng(seed + 7919).permutation(num_nodes // 2).astype(np.int64)
            u, v = 2 * rp[u >> 1] + (u & 1), 2 * rp[v >> 1] + (v & 1)
    elif kind == "uniform":
        u = rng.integers(0, num_nodes, size=half)
        v = rng.integers(0, num_nodes, size=half)
    else:
        raise ValueError(kind)
    src = np.concatenate([u, v ^ 1])
    dst = np.concatenate([v, u ^ 1])
    ol_len = rng.standard_normal(half).astype(np.float32)
    ol_sim = rng.uniform(0.9, 1.0, size=half).astype(np.float32)
    feat = np.stack([np.concatenate([ol_len, ol_len]), np.concatenate([ol_sim, ol_sim])], axis=1)
    yy = (rng.random(half) < 0.3).astype(np.float32)
    y = np.concatenate([yy, yy])
    perm = rng.permutation(num_edges)
    return {
        "src": torch.from_numpy(src[perm].astype(np.int32)),
        "dst": torch.from_numpy(dst[perm].astype(np.int32)),
        "num_nodes": int(num_nodes),
        "e": torch.from_numpy(np.ascontiguousarray(feat[perm])),
        "y": torch.from_numpy(y[perm]),
        "pos_weight": torch.tensor(0.7 / 0.3),
    }


def random_state_dict(hidden, num_layers=8, hidden_ne=16, hidden_edge_scores=64, node_features=2,
                      edge_features=2, seed=1):
    """Default nn.Linear / BatchNorm1d init under manual_seed(seed), then non-trivial BN buffers
    (running_mean ~ N(0, 0.5^2), running_var ~ U(0.05, 1)) so eval-mode BN is exercised."""
    import torch.nn as nn
    g = torch.Generator().manual_seed(seed)
    sd = {}

    def linear(name, fan_in, fan_out):
        bound = 1.0 / fan_in ** 0.5
        sd[name + ".weight"] = (torch.rand(fan_out, fan_in, generator=g) * 2 - 1) * bound
        sd[name + ".bias"] = (torch.rand(fan_out, generator=g) * 2 - 1) * bound

    linear("linear1_node", node_features, hidden_ne)
    linear("linear2_node", hidden_ne, hidden)
    linear("linear1_edge", edge_features, hidden_ne)
    linear("linear2_edge", hidden_ne, hidden)
    for l in range(num_layers):
        p = f"gnn.convs.{l}."
        for nm in ("A_1", "A_2", "A_3", "B_1", "B_2", "B_3"):
            linear(p + nm, hidden, hidden)
        for bn in ("bn_h", "bn_e"):
            sd[p + bn + ".weight"] = 1.0 + 0.1 * torch.randn(hidden, generator=g)
            sd[p + bn + ".bias"] = 0.1 * torch.randn(hidden, generator=g)
            sd[p + bn + ".running_mean"] = 0.5 * torch.randn(hidden, generator=g)
            sd[p + bn + ".running_var"] = 0.05 + 0.95 * torch.rand(hidden, generator=g)
            sd[p + bn + ".num_batches_tracked"] = torch.tensor(0, dtype=torch.int64)
    linear("predictor.W1", 3 * hidden, hidden_edge_scores)
    linear("predictor.W2", hidden_edge_scores, 32)
    linear("predictor.W3", 32, 1)
    del nn
    return sd
