#!/usr/bin/env python
"""Training step (configs[2] shape) with the gate's forward as two passes (statistics alone, then gate + xe out; round 4, the default)
against the three-pass form (raw gate + statistics, bn_relu_res): hipGraph-free eager steps, same process, alternating.
usage: tools/train_two_pass_ab.py [fp32|bf16]"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnnome_amd  # noqa: E402
from gnnome_amd import ops, train  # noqa: E402
from gnnome_amd.synth import make_graph, random_state_dict  # noqa: E402

dev = torch.device("cuda", 0)
n, e, H = 100_000, 1_000_000, 128
g = make_graph(n, e, seed=1)
views = ops.GraphViews(g["src"].to(dev), g["dst"].to(dev), n)
x, ef, y, pw = ops.degree_features(views), g["e"].to(dev), g["y"].to(dev), g["pos_weight"].to(dev)
ref = None
for rnd in range(3):
    for two in (True, False):
        train.TWO_PASS_GATE = two
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
        print(f"round {rnd} two-pass gate {two}: {s.elapsed_time(t) / 10:.2f} ms / step (eager, no optimizer), loss {loss.item():.7f} (first {ref[0]:.7f}), "
              f"|grad| {gn:.6e} (first {ref[1]:.6e})", flush=True)
train.TWO_PASS_GATE = True
