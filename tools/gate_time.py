#!/usr/bin/env python
"""Median launch time of the edge-gate kernel per variant at a BASELINE size (HIP events, back-to-back launches
with a double-buffered e so the input is never the just-written output)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnnome_amd import ops  # noqa: E402
from gnnome_amd.synth import make_graph  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--variants", default="0,5")
ap.add_argument("--hidden", type=int, default=128)
ap.add_argument("--edges", type=int, default=1_000_000)
ap.add_argument("--reps", type=int, default=30)
ap.add_argument("--xps", default="0", help="comma list of gnnome_set_tuning(4, x) experiment masks")
ap.add_argument("--ablations", default="0", help="comma list of gnnome_set_tuning(1, mask) values; results are only compared for mask 0")
a = ap.parse_args()
dev = torch.device("cuda", 0)
n, e, H = a.edges // 10, a.edges, a.hidden
g = make_graph(n, e, seed=1)
views = ops.GraphViews(g["src"].to(dev), g["dst"].to(dev), n)
gen = torch.Generator(device=dev).manual_seed(0)
ee = torch.randn(e, H, device=dev, generator=gen)
out = torch.empty_like(ee)
P = torch.randn(n, 5 * H, device=dev, generator=gen)
W3 = torch.randn(H, H, device=dev, generator=gen) / H ** 0.5
sc, sh = torch.rand(H, device=dev, generator=gen) * 0.1, torch.randn(H, device=dev, generator=gen)
ref = None
for rnd in range(2):
  for abl, xp in ((int(t), int(x)) for t in a.ablations.split(",") for x in a.xps.split(",")):
    ops.set_tuning(1, abl)
    ops.set_tuning(4, xp)
    for v in (int(t) for t in a.variants.split(",")):
        ops.set_tuning(0, v)
        for _ in range(3):
            ops.edge_gate(ee, P[:, 3 * H:4 * H], P[:, 4 * H:], views, W3, 0, sc, sh, out=out)
        evs = []
        for _ in range(a.reps):
            s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            ops.edge_gate(ee, P[:, 3 * H:4 * H], P[:, 4 * H:], views, W3, 0, sc, sh, out=out)
            t.record()
            evs.append((s, t))
        torch.cuda.synchronize()
        ts = sorted(x.elapsed_time(y) for x, y in evs)
        if ref is None and abl == 0:
            ref = out.clone()
        same = torch.equal(ref, out) if abl == 0 else None
        print(f"round {rnd} ablation {abl:2d} xp {xp} variant {v}: median {ts[len(ts) // 2]:.4f} ms  min {ts[0]:.4f} ms  "
              f"{2.0 * e * H * H / (ts[len(ts) // 2] * 1e-3) / 1e12:.1f} TF  bit-identical to first variant: {same}", flush=True)
ops.set_tuning(0, 0)
ops.set_tuning(1, 0)
ops.set_tuning(4, 0)
