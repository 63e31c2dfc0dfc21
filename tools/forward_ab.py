#!/usr/bin/env python
"""Whole forward (BASELINE configs[1]) per gate variant (gnnome_set_tuning key 0): 0 = default (plane form of the
edge-tile kernel at H = 128), 8 = second-generation kernel everywhere."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnnome_amd  # noqa: E402
from gnnome_amd import ops  # noqa: E402
from gnnome_amd.synth import make_graph, random_state_dict  # noqa: E402

dev = torch.device("cuda", 0)
n, e, H = 100_000, 1_000_000, 128
g = make_graph(n, e, seed=1)
m = gnnome_amd.SymGatedGCNModel(2, 2, H, 16, 8, 64, "batch").eval()
m.load_state_dict(random_state_dict(H, seed=1))
m.to(dev)
views = ops.GraphViews(g["src"].to(dev), g["dst"].to(dev), n)
x, ef = ops.degree_features(views), g["e"].to(dev)
ref = None
KEY = int(sys.argv[2]) if len(sys.argv) > 2 else 0   # tuning key the variants belong to (0 = gate variant, 7 = aggregation variant)
with torch.no_grad():
    for rnd in range(2):
        for v in [int(t) for t in (sys.argv[1] if len(sys.argv) > 1 else "0,8").split(",")]:
            ops.set_tuning(KEY, v)
            for _ in range(5):
                out = m(views, x, ef)
            s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(50):
                out = m(views, x, ef)
            t.record()
            torch.cuda.synchronize()
            ref = out if ref is None else ref
            print(f"round {rnd} key {KEY} variant {v}: {s.elapsed_time(t) / 50:.3f} ms / forward, max|dp| vs first {(torch.sigmoid(out) - torch.sigmoid(ref)).abs().max().item():.2e}", flush=True)
ops.set_tuning(KEY, 0)
