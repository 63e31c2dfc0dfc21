#!/usr/bin/env python
"""Race hunt for the fp16x3 + LDS-DMA edge-tile kernel at full size: the same launch many times, every result compared bit for bit with the first
(a lost or early hand-over shows as a handful of differing rows per million edges).  usage: tools/f16_repeat.py [edges] [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnnome_amd import ops  # noqa: E402
from gnnome_amd.synth import make_graph  # noqa: E402

dev = torch.device("cuda", 0)
e = int(sys.argv[1]) if len(sys.argv) > 1 else 2_500_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
n, H = e // 10, 256
g = make_graph(n, e, seed=1)
views = ops.GraphViews(g["src"].to(dev), g["dst"].to(dev), n)
gen = torch.Generator(device=dev).manual_seed(0)
ee = torch.randn(e, H, device=dev, generator=gen)
P = torch.randn(n, 5 * H, device=dev, generator=gen)
W3 = torch.randn(H, H, device=dev, generator=gen) / H ** 0.5
Wc, bc = torch.randn(5 * H, H, device=dev, generator=gen) / 16, torch.randn(5 * H, device=dev, generator=gen)
sc, sh = torch.rand(H, device=dev, generator=gen) * 0.1, torch.randn(H, device=dev, generator=gen)
enc = (torch.randn(16, 2, device=dev, generator=gen), torch.randn(16, device=dev, generator=gen), torch.randn(H, 16, device=dev, generator=gen) / 4,
       torch.randn(H, device=dev, generator=gen))
e_raw = g["e"].to(dev)
hrows = torch.randn(n, H, device=dev, generator=gen)
runs = {
    "gate (mode 0)": lambda: ops.edge_gate(ee, P[:, 3 * H:4 * H], P[:, 4 * H:], views, W3, 0, sc, sh, out=torch.empty_like(ee)),
    "raw gate + statistics (mode 1)": lambda: torch.cat([t.flatten() for t in ops.edge_gate_raw_stats(ee, P[:, 3 * H:4 * H], P[:, 4 * H:], views, W3)]),
    "projection (mode 4)": lambda: ops.linear(hrows, Wc, bc),
    "folded-encoder gate (mode 5)": lambda: ops.edge_gate_encode(e_raw, enc, P[:, 3 * H:4 * H], P[:, 4 * H:], views, W3, sc, sh),
}
for name, fn in runs.items():
    first = fn().clone()
    bad = 0
    for _ in range(reps):
        out = fn()
        if not torch.equal(out, first):
            bad += 1
    torch.cuda.synchronize()
    print(f"{name}: {reps} launches at E = {e}, {bad} differ from the first, finite: {bool(torch.isfinite(first).all())}", flush=True)
    assert bad == 0
print("repeat ok")
