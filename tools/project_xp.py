#!/usr/bin/env python
"""Node projection (csrc/node_project.hip): the share of a CU's units given to its first-dispatched workgroup (gnnome_set_tuning key 4 = percent;
0 = the default 55, 50 = equal runs), alternating in one process."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnnome_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
H = int(sys.argv[1]) if len(sys.argv) > 1 else 128
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000
xps = [int(v) for v in (sys.argv[3].split(",") if len(sys.argv) > 3 else "0,50,52,58,60,64".split(","))]
gen = torch.Generator(device=dev).manual_seed(0)
h = torch.randn(n, H, device=dev, generator=gen)
W = torch.randn(5 * H, H, device=dev, generator=gen) / H ** 0.5
b = torch.randn(5 * H, device=dev, generator=gen)
out = torch.empty(n, 5 * H, device=dev)
planes = ops.weight_planes(W)
ref = ops.linear(h, W, b, planes=planes).clone()
for rnd in range(3):
    for xp in xps:
        ops.set_tuning(4, xp)
        for _ in range(5):
            ops.linear(h, W, b, out=out, planes=planes)
        evs = []
        for _ in range(40):
            s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            ops.linear(h, W, b, out=out, planes=planes)
            t.record()
            evs.append((s, t))
        torch.cuda.synchronize()
        ts = sorted(x.elapsed_time(y) for x, y in evs)
        print(f"round {rnd} H={H} N={n} xp={xp:5d}: median {ts[len(ts) // 2] * 1e3:.1f} us  min {ts[0] * 1e3:.1f} us  same bits {bool(torch.equal(out, ref))}", flush=True)
ops.set_tuning(4, 0)
