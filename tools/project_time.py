#!/usr/bin/env python
"""Round 6: the node projection [N,H] x [H,5H] on gnnome_linear_planes_f32 (csrc/node_project.hip) against rounds 4-5's fp16x3 edge-tile
kernels (gnnome_set_tuning(2, 10)), alternating in one process, and the new kernel's measurement-only probes (key 2 = 20, key 1 = mask:
1 no stores, 2 no MFMAs, 4 no DMA in the loop, 8 no quad transpose, 16 no row prefetch).  `project_time.py 128 100000` / `project_time.py 256 253000`."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnnome_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
H = int(sys.argv[1]) if len(sys.argv) > 1 else 128
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000
arms = sys.argv[3].split(",") if len(sys.argv) > 3 else ["new", "old", "p1", "p2", "p4", "p3"]
gen = torch.Generator(device=dev).manual_seed(0)
h = torch.randn(n, H, device=dev, generator=gen)
W = torch.randn(5 * H, H, device=dev, generator=gen) / H ** 0.5
b = torch.randn(5 * H, device=dev, generator=gen)
out = torch.empty(n, 5 * H, device=dev)
planes = ops.weight_planes(W)
want = None


def select(arm):
    ops.set_tuning(1, 0)
    if arm == "new":
        ops.set_tuning(2, 0)
    elif arm == "old":
        ops.set_tuning(2, 10)
    else:
        ops.set_tuning(2, 20)
        ops.set_tuning(1, int(arm[1:]))


for rnd in range(3):
    for arm in arms:
        select(arm)
        call = (lambda: ops.linear(h, W, b, out=out)) if arm == "old" else (lambda: ops.linear(h, W, b, out=out, planes=planes))
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
        note = ""
        if arm in ("new", "old"):
            if want is None:
                want = (h[:4096].double() @ W.double().t() + b.double())
            note = f"  max err vs fp64 (first 4096 rows) {(out[:4096].double() - want).abs().max().item():.2e}"
        gb = (n * H * 4 + n * 5 * H * 4) / 1e9
        print(f"round {rnd} H={H} N={n} {arm:>4}: median {ts[len(ts) // 2] * 1e3:.1f} us  min {ts[0] * 1e3:.1f} us  ({gb / ts[len(ts) // 2] :.2f} TB/s algorithmic){note}", flush=True)
ops.set_tuning(2, 0)
ops.set_tuning(1, 0)
