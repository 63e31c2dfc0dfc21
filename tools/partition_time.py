#!/usr/bin/env python
"""Wall time of the clusters for mini-batch training (gnnome_amd.partition.cluster_partition = train.py:333-336's dgl.metis_partition(g, N / 2000,
extra_cached_hops=1), which the reference calls every epoch): the multilevel k-way labelling, the sub-graphs with their halo, and the cut."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnnome_amd import partition  # noqa: E402
from gnnome_amd.synth import make_graph  # noqa: E402

dev = torch.device("cuda", 0)
for n in [int(v) for v in (sys.argv[1:] or ["100000", "1000000"])]:
    e, k = 10 * n, max(2, n // 2000)     # hyperparameters.py:36: num_nodes_per_cluster = 2000
    g = make_graph(n, e, seed=1)
    src, dst = g["src"].to(dev), g["dst"].to(dev)
    for rnd in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        label = partition.multilevel_partition(src, dst, n, k)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        parts = partition.cluster_partition((src, dst, n), k, extra_cached_hops=1, device=dev)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        sizes = torch.bincount(label, minlength=k)
        cut = partition.edge_cut(src, dst, label)
        ranges = torch.clamp(torch.arange(n, device=dev) * k // n, max=k - 1)
        print(f"N={n} E={e} k={k} run {rnd}: labelling {1e3 * (t1 - t0):.0f} ms; cluster_partition (labelling + {len(parts)} sub-graphs with one-hop halo and "
              f"their views) {1e3 * (t2 - t1):.0f} ms; cut {cut} = {cut / e:.2%} of the edges (contiguous ranges: {partition.edge_cut(src, dst, ranges)}), "
              f"largest part {int(sizes.max())} of {n / k:.0f} + 3 %", flush=True)
