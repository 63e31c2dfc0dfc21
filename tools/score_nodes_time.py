#!/usr/bin/env python
"""Round 6: the scorer's node halves [N,H] x [H,2 hs] (score_predictor.py:13-14) on the planes kernel (gnnome_linear_planes_f32, planes given) against the
kernel ops.linear picks for that shape by itself (bf16x6 at 2 hs < 256 with K = 128).  `score_nodes_time.py 128 100000 64`."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnnome_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
H = int(sys.argv[1]) if len(sys.argv) > 1 else 128
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000
hs = int(sys.argv[3]) if len(sys.argv) > 3 else 64
gen = torch.Generator(device=dev).manual_seed(0)
h = torch.randn(n, H, device=dev, generator=gen)
W = torch.randn(2 * hs, H, device=dev, generator=gen) / H ** 0.5
b = torch.randn(2 * hs, device=dev, generator=gen)
out = torch.empty(n, 2 * hs, device=dev)
planes = ops.weight_planes(W)
want = h[:4096].double() @ W.double().t() + b.double()
for rnd in range(3):
    for arm, call in (("by itself", lambda: ops.linear(h, W, b, out=out)), ("planes", lambda: ops.linear(h, W, b, out=out, planes=planes))):
        for _ in range(5):
            call()
        evs = []
        for _ in range(40):
            s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            call()
            t.record()
            evs.append((s, t))
        torch.cuda.synchronize()
        ts = sorted(x.elapsed_time(y) for x, y in evs)
        err = (out[:4096].double() - want).abs().max().item()
        print(f"round {rnd} H={H} N={n} 2hs={2 * hs} {arm:>9}: median {ts[len(ts) // 2] * 1e3:.1f} us  min {ts[0] * 1e3:.1f} us  max err vs fp64 {err:.2e}", flush=True)
