#!/usr/bin/env python
"""Launch time of the score predictor's tail backward (gnnome_score_tail_bwd_f32): tiled kernel against the row-per-lane one (key 4 = 79)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnnome_amd import ops  # noqa: E402
from gnnome_amd.synth import make_graph  # noqa: E402

dev = torch.device("cuda", 0)
e = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
hs = 64
g = make_graph(e // 10, e, seed=1)
views = ops.GraphViews(g["src"].to(dev), g["dst"].to(dev), e // 10)
gen = torch.Generator(device=dev).manual_seed(0)
z1 = torch.relu(torch.randn(e, hs, device=dev, generator=gen))
ds = torch.randn(e, device=dev, generator=gen)
W2, b2, W3 = torch.randn(32, hs, device=dev, generator=gen) / 8, torch.randn(32, device=dev, generator=gen), torch.randn(32, device=dev, generator=gen)
for rnd in range(2):
    for xp in (0, 79):
        ops.set_tuning(4, xp)
        for _ in range(2):
            ops.score_tail_bwd(z1, ds, views, W2, b2, W3)
        evs = []
        for _ in range(10):
            s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            ops.score_tail_bwd(z1, ds, views, W2, b2, W3)
            t.record()
            evs.append((s, t))
        torch.cuda.synchronize()
        ts = sorted(x.elapsed_time(y) for x, y in evs)
        print(f"round {rnd} kernel {'row per lane' if xp else 'tiles'}: median {ts[len(ts) // 2]:.4f} ms  ({e * (2 * hs + 64) * 4 / ts[len(ts) // 2] / 1e9:.2f} TB/s)", flush=True)
ops.set_tuning(4, 0)
