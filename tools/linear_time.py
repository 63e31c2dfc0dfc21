#!/usr/bin/env python
"""Launch time of the node projection [N,H] x [H,5H] (gnnome_linear_f32) per kernel variant (gnnome_set_tuning key 2, or the key given as the
fourth argument: `linear_time.py 128 100000 0,1 11` = stores from the compute waves against the x-tile route)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnnome_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
H = int(sys.argv[1]) if len(sys.argv) > 1 else 128
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000
variants = [int(v) for v in (sys.argv[3].split(",") if len(sys.argv) > 3 else "0,4,5,3,2".split(","))]
key = int(sys.argv[4]) if len(sys.argv) > 4 else 2
gen = torch.Generator(device=dev).manual_seed(0)
h = torch.randn(n, H, device=dev, generator=gen)
W = torch.randn(5 * H, H, device=dev, generator=gen)
b = torch.randn(5 * H, device=dev, generator=gen)
out = torch.empty(n, 5 * H, device=dev)
ref = None
for rnd in range(3):
    for v in variants:
        ops.set_tuning(key, v)
        for _ in range(3):
            ops.linear(h, W, b, out=out)
        evs = []
        for _ in range(30):
            s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            ops.linear(h, W, b, out=out)
            t.record()
            evs.append((s, t))
        torch.cuda.synchronize()
        ts = sorted(x.elapsed_time(y) for x, y in evs)
        if ref is None:
            ref = out.clone()
        print(f"round {rnd} variant {v}: median {ts[len(ts) // 2]:.4f} ms  min {ts[0]:.4f} ms  max|diff to variant {variants[0]}| {(out - ref).abs().max().item():.2e}", flush=True)
ops.set_tuning(key, 0)
