#!/usr/bin/env python
"""Whole forward (configs[1]) against the aggregation's items in flight per lane group (gnnome_set_tuning key 7) x its resident
workgroups per CU (key 5: KiB of unused dynamic LDS per workgroup): the window of nodes in flight against latency hiding."""
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
with torch.no_grad():
    for rnd in range(2):
        for variant in (0, 3):
            for kib in (0, 20, 26, 32, 40):
                ops.set_tuning(7, variant)
                ops.set_tuning(5, kib)
                for _ in range(5):
                    m(views, x, ef)
                s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(50):
                    m(views, x, ef)
                t.record()
                torch.cuda.synchronize()
                print(f"round {rnd} aggregation variant {variant} LDS cap {kib} KiB: {s.elapsed_time(t) / 50:.3f} ms / forward", flush=True)
ops.set_tuning(7, 0)
ops.set_tuning(5, 0)
