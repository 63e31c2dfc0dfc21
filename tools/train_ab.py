#!/usr/bin/env python
"""Training step (configs[2] shape, eager) per gate variant (gnnome_set_tuning key 0): 0 = plane form of the edge-tile kernel in
modes 1 and 3, 8 = second generation."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnnome_amd  # noqa: E402
from gnnome_amd import ops  # noqa: E402
from gnnome_amd.synth import make_graph, random_state_dict  # noqa: E402

dev = torch.device("cuda", 0)
n, e, H = 100_000, 1_000_000, 128
g = make_graph(n, e, seed=1)
views = ops.GraphViews(g["src"].to(dev), g["dst"].to(dev), n)
x, ef, y, pw = ops.degree_features(views), g["e"].to(dev), g["y"].to(dev), g["pos_weight"].to(dev)
ref = None
for rnd in range(2):
    for v in (0, 8):
        ops.set_tuning(0, v)
        m = gnnome_amd.SymGatedGCNModel(2, 2, H, 16, 8, 64, "batch").train()
        m.load_state_dict(random_state_dict(H, seed=1))
        m.to(dev)
        if len(sys.argv) > 1:
            m.activation_storage = sys.argv[1]

        def step():
            m.zero_grad(set_to_none=True)
            loss = F.binary_cross_entropy_with_logits(m(views, x, ef).squeeze(-1), y, pos_weight=pw)
            loss.backward()
            return loss

        for _ in range(3):
            loss = step()
        s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10):
            loss = step()
        t.record()
        torch.cuda.synchronize()
        gn = sum(p.grad.double().pow(2).sum().item() for p in m.parameters()) ** 0.5
        ref = (loss.item(), gn) if ref is None else ref
        print(f"round {rnd} gate variant {v}: {s.elapsed_time(t) / 10:.2f} ms / step (eager, no optimizer), loss {loss.item():.7f} (first {ref[0]:.7f}), |grad| {gn:.6e} (first {ref[1]:.6e})", flush=True)
ops.set_tuning(0, 0)
