#!/usr/bin/env python
"""Cost of a FRESH graph in a warm process: GraphViews (two radix sorts + CSR pointers) + degree features for new edge lists,
with a model forward in between (the allocator state a real loop has)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnnome_amd  # noqa: E402
from gnnome_amd import ops  # noqa: E402
from gnnome_amd.synth import make_graph, random_state_dict  # noqa: E402

dev = torch.device("cuda", 0)
n, e, H = 100_000, 1_000_000, 128
m = gnnome_amd.SymGatedGCNModel(2, 2, H, 16, 8, 64, "batch").eval()
m.load_state_dict(random_state_dict(H, seed=1))
m.to(dev)
for i in range(6):
    g = make_graph(n, e, seed=10 + i)
    src, dst, ef = g["src"].to(dev), g["dst"].to(dev), g["e"].to(dev)
    if i == 3:
        src, dst = dst, src     # the reversed list (what bench.py's cold record rebuilds)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    v = ops.GraphViews(src, dst, n, validate="lazy")
    t1 = time.perf_counter()
    x = ops.degree_features(v)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    with torch.no_grad():
        m(v, x, ef)
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    print(f"graph {i}: views {1e3 * (t1 - t0):.2f} ms (host side) + features, synchronised {1e3 * (t2 - t0):.2f} ms; forward incl. host {1e3 * (t3 - t2):.2f} ms", flush=True)
