#!/usr/bin/env python
"""Which torch operators the training step still launches (per step, with the calling line in gnnome_amd/): the census behind
the small-launch clean-up of DESIGN 4b."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnnome_amd  # noqa: E402
from gnnome_amd import ops  # noqa: E402
from gnnome_amd.loss import bce_loss  # noqa: E402
from gnnome_amd.synth import make_graph, random_state_dict  # noqa: E402

dev = torch.device("cuda", 0)
n, e = 4000, 40000
g = make_graph(n, e, seed=1)
views = ops.GraphViews(g["src"].to(dev), g["dst"].to(dev), n)
x = ops.degree_features(views)
ef, y, pw = g["e"].to(dev), g["y"].to(dev), g["pos_weight"].to(dev)
m = gnnome_amd.SymGatedGCNModel(2, 2, 128, 16, 8, 64, "batch")
m.load_state_dict(random_state_dict(128, seed=1))
m.to(dev).train()


def step():
    logits = m(views, x, ef)
    loss = bce_loss(logits.squeeze(-1), y, pw)
    m.zero_grad(set_to_none=False)
    loss.backward()


for _ in range(3):
    step()
torch.cuda.synchronize()
steps = 2
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU], with_stack=True) as prof:
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
rows = {}
for ev in prof.events():
    if not ev.name.startswith("aten::") or ev.cpu_parent is not None and ev.cpu_parent.name.startswith("aten::"):
        continue
    site = next((f for f in ev.stack if "gnnome_amd/" in f and "_call" not in f), ev.stack[0] if ev.stack else "?")
    key = (ev.name, site.split("gnnome_amd/")[-1][:70])
    rows[key] = rows.get(key, 0) + 1
total = 0
for (name, site), c in sorted(rows.items(), key=lambda kv: -kv[1]):
    total += c
    print(f"{c / steps:7.1f}  {name:28s} {site}")
print(f"{total / steps:7.1f}  top-level aten operators per step")
