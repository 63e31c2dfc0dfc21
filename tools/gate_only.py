#!/usr/bin/env python
"""Run only the edge-gate kernel (chosen variant) a few times at a BASELINE size - a small target for rocprofv3 --pmc."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnnome_amd import ops  # noqa: E402
from gnnome_amd.synth import make_graph  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--variant", type=int, default=0)
ap.add_argument("--ablation", type=int, default=0)
ap.add_argument("--hidden", type=int, default=128)
ap.add_argument("--edges", type=int, default=1_000_000)
ap.add_argument("--reps", type=int, default=5)
a = ap.parse_args()
dev = torch.device("cuda", 0)
n, e, H = a.edges // 10, a.edges, a.hidden
g = make_graph(n, e, seed=1)
views = ops.GraphViews(g["src"].to(dev), g["dst"].to(dev), n)
gen = torch.Generator(device=dev).manual_seed(0)
ee = torch.randn(e, H, device=dev, generator=gen)
P = torch.randn(n, 5 * H, device=dev, generator=gen)
W3 = torch.randn(H, H, device=dev, generator=gen) / H ** 0.5
sc, sh = torch.rand(H, device=dev, generator=gen) * 0.1, torch.randn(H, device=dev, generator=gen)
ops.set_tuning(0, a.variant)
ops.set_tuning(1, a.ablation)
out = torch.empty_like(ee) if H == 256 else None   # (the H = 256 edge-tile kernel wants separate buffers: in place runs another kernel)
for _ in range(a.reps):
    ops.edge_gate(ee, P[:, 3 * H:4 * H], P[:, 4 * H:], views, W3, 0, sc, sh, out=out)
torch.cuda.synchronize()
