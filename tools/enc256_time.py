#!/usr/bin/env python
"""The folded-encoder gate at H = 256 (k_edge_tile_f16<5>: an epilogue-only kernel - gathers + stores) with its gathers / its stores compiled out
(gnnome_set_tuning(1, 308 / 316), wrong results): what the epilogue waves' two kinds of traffic cost on their own.
`enc256_time.py 128 [edges]`: the launch time of the H = 128 form (k_edge_gate_enc16; no ablations there)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnnome_amd import ops  # noqa: E402
from gnnome_amd.synth import make_graph  # noqa: E402

dev = torch.device("cuda", 0)
H = int(sys.argv[1]) if len(sys.argv) > 1 else 256
e = int(sys.argv[2]) if len(sys.argv) > 2 else (2_500_000 if H == 256 else 1_000_000)
n = e // 10
g = make_graph(n, e, seed=1)
views = ops.GraphViews(g["src"].to(dev), g["dst"].to(dev), n)
gen = torch.Generator(device=dev).manual_seed(0)
P = torch.randn(n, 5 * H, device=dev, generator=gen)
W3 = torch.randn(H, H, device=dev, generator=gen) / H ** 0.5
sc, sh = torch.rand(H, device=dev, generator=gen) * 0.1, torch.randn(H, device=dev, generator=gen)
enc = (torch.randn(16, 2, device=dev, generator=gen), torch.randn(16, device=dev, generator=gen), torch.randn(H, 16, device=dev, generator=gen) / 4,
       torch.randn(H, device=dev, generator=gen))
e_raw = g["e"].to(dev)
for rnd in range(2):
    for abl in ((0, 308, 316) if H == 256 else (0,)):
        ops.set_tuning(1, abl)
        for _ in range(3):
            ops.edge_gate_encode(e_raw, enc, P[:, 3 * H:4 * H], P[:, 4 * H:], views, W3, sc, sh)
        evs = []
        for _ in range(20):
            s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            ops.edge_gate_encode(e_raw, enc, P[:, 3 * H:4 * H], P[:, 4 * H:], views, W3, sc, sh)
            t.record()
            evs.append((s, t))
        torch.cuda.synchronize()
        ts = sorted(x.elapsed_time(y) for x, y in evs)
        print(f"H={H} E={e} round {rnd} ablation {abl}: median {ts[len(ts) // 2]:.4f} ms  min {ts[0]:.4f} ms", flush=True)
ops.set_tuning(1, 0)
