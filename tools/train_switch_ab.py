#!/usr/bin/env python
"""Training step (configs[2] shape, eager, no optimizer) with one of gnnome_amd.train's module switches on and off, alternating:
python tools/train_switch_ab.py FUSED_AGG_BWD [fp32|bf16] [H] [nodes] [edges]   (also TWO_PASS_GATE)"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnnome_amd  # noqa: E402
import gnnome_amd.train as train  # noqa: E402
from gnnome_amd import ops  # noqa: E402
from gnnome_amd.synth import make_graph, random_state_dict  # noqa: E402

dev = torch.device("cuda", 0)
switch = sys.argv[1] if len(sys.argv) > 1 else "FUSED_AGG_BWD"
storage = sys.argv[2] if len(sys.argv) > 2 else "fp32"
H = int(sys.argv[3]) if len(sys.argv) > 3 else 128
n = int(sys.argv[4]) if len(sys.argv) > 4 else 100_000
e = int(sys.argv[5]) if len(sys.argv) > 5 else 1_000_000
assert hasattr(train, switch), switch
g = make_graph(n, e, seed=1)
views = ops.GraphViews(g["src"].to(dev), g["dst"].to(dev), n)
x, ef, y, pw = ops.degree_features(views), g["e"].to(dev), g["y"].to(dev), g["pos_weight"].to(dev)
m = gnnome_amd.SymGatedGCNModel(2, 2, H, 16, 8, 64, "batch").train()
m.load_state_dict(random_state_dict(H, seed=1))
m.to(dev)
m.activation_storage = storage


def step():
    m.zero_grad(set_to_none=True)
    loss = F.binary_cross_entropy_with_logits(m(views, x, ef).squeeze(-1), y, pos_weight=pw)
    loss.backward()
    return loss


ref = {}
for rnd in range(3):
    for on in (True, False):
        setattr(train, switch, on)
        for _ in range(2):
            loss = step()
        s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10):
            loss = step()
        t.record()
        torch.cuda.synchronize()
        grads = torch.cat([p.grad.double().reshape(-1) for p in m.parameters()])
        ref.setdefault(on, grads)
        other = ref.get(not on)
        rel = "" if other is None else f", gradient L2 distance to the other setting {((grads - other).norm() / other.norm()).item():.2e}"
        print(f"round {rnd} {switch}={on}: {s.elapsed_time(t) / 10:.3f} ms / step (eager, no optimizer, {storage} storage), loss {loss.item():.7f}{rel}",
              flush=True)
setattr(train, switch, True)
