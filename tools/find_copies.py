#!/usr/bin/env python
"""Which torch operators does one eval forward issue besides the library's launches?  (torch.profiler, E. coli-sized graph)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnnome_amd  # noqa: E402
from gnnome_amd import ops  # noqa: E402
from gnnome_amd.synth import make_graph  # noqa: E402

dev = torch.device("cuda", 0)
n, e = 30_000, 300_000
g = make_graph(n, e, seed=1)
m = gnnome_amd.SymGatedGCNModel(2, 2, 64, 16, 8, 64, "batch").eval()
m.load_state_dict(torch.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "weights.pt"), map_location="cpu"))
m.to(dev)
views = ops.GraphViews(g["src"].to(dev), g["dst"].to(dev), n)
x, ef = ops.degree_features(views), g["e"].to(dev)
for _ in range(3):
    m(views, x, ef)
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile  # noqa: E402
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for _ in range(5):
        m(views, x, ef)
    torch.cuda.synchronize()
print(prof.key_averages(group_by_stack_n=6).table(sort_by="count", row_limit=25, max_name_column_width=60))
