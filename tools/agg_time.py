#!/usr/bin/env python
"""Launch time of the aggregation kernel against the resident-workgroup cap (gnnome_set_tuning key 5) at a BASELINE size."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnnome_amd import ops  # noqa: E402
from gnnome_amd.synth import make_graph  # noqa: E402

dev = torch.device("cuda", 0)
H = int(sys.argv[1]) if len(sys.argv) > 1 else 128
n, e = 100_000, 1_000_000
g = make_graph(n, e, seed=1)
views = ops.GraphViews(g["src"].to(dev), g["dst"].to(dev), n)
gen = torch.Generator(device=dev).manual_seed(0)
ee = torch.randn(e, H, device=dev, generator=gen)
h = torch.randn(n, H, device=dev, generator=gen)
P = torch.randn(n, 5 * H, device=dev, generator=gen)
sc, sh = torch.rand(H, device=dev, generator=gen) * 0.1, torch.randn(H, device=dev, generator=gen)
A1, A2, A3 = (P[:, i * H:(i + 1) * H] for i in range(3))
if len(sys.argv) > 2 and sys.argv[2] == "once":   # a few launches of the default kernel, for the PMC passes (tools/pmc_agg.sh)
    for _ in range(5):
        ops.node_aggregate(ee, A1, A2, A3, views, h, 0, sc, sh)
    torch.cuda.synchronize()
    sys.exit(0)
VARIANTS = {0: "default (U=4 at H<=128, 8 at 256)", 1: "U=4, 8 waves/SIMD", 2: "U=2, 8 waves/SIMD", 3: "U=1", 4: "U=8", 5: "U=2", 6: "U=4, single item loop (in-edge rows requested after the index wait; the round-1 form)",
            9: "two nodes per wave, U=4 (k_node_aggregate_pair, round 4)", 10: "two nodes per wave, U=2", 11: "two nodes per wave, persistent contiguous chunks, 4 WG/CU", 12: "... + nontemporal e loads", 13: "... 8 WG/CU", 14: "default + nontemporal out-edge loads of e' (round 5)", 15: "MEASUREMENT ONLY: one L1-resident table row per node (no gather misses; round 6)", 7: "MEASUREMENT ONLY: out-edges alone"}
if len(sys.argv) > 3:   # a subset: python tools/agg_time.py 128 variants 0,14
    VARIANTS = {int(v): VARIANTS[int(v)] for v in sys.argv[3].split(",")}
if len(sys.argv) > 2 and sys.argv[2] == "variants":   # items in flight per lane group against occupancy (gnnome_set_tuning key 7)
    for rnd in range(3):
        for v, name in VARIANTS.items():
            ops.set_tuning(7, v)
            for _ in range(3):
                ops.node_aggregate(ee, A1, A2, A3, views, h, 0, sc, sh)
            evs = []
            for _ in range(30):
                s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                ops.node_aggregate(ee, A1, A2, A3, views, h, 0, sc, sh)
                t.record()
                evs.append((s, t))
            torch.cuda.synchronize()
            ts = sorted(x.elapsed_time(y) for x, y in evs)
            print(f"round {rnd} variant {v} ({name}): median {ts[len(ts) // 2]:.4f} ms  min {ts[0]:.4f} ms", flush=True)
    ops.set_tuning(7, 0)
    sys.exit(0)
for rnd in range(2):
    for kib in (0, 16, 24, 32, 40, 54, 80):
        ops.set_tuning(5, kib)
        for _ in range(3):
            ops.node_aggregate(ee, A1, A2, A3, views, h, 0, sc, sh)
        evs = []
        for _ in range(30):
            s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            ops.node_aggregate(ee, A1, A2, A3, views, h, 0, sc, sh)
            t.record()
            evs.append((s, t))
        torch.cuda.synchronize()
        ts = sorted(x.elapsed_time(y) for x, y in evs)
        print(f"round {rnd} dynamic LDS {kib:3d} KiB (<= {160 // max(kib, 1) if kib else 8} workgroups/CU): median {ts[len(ts) // 2]:.4f} ms  min {ts[0]:.4f} ms", flush=True)
ops.set_tuning(5, 0)
