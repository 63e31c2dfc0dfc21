#!/usr/bin/env python
"""In-kernel cycle counters of the node-projection kernel (csrc/node_project.hip, PROBE 32): per workgroup the prologue (requests for h, W's
first chunks, bias -> landed -> split), the chunk loop, the part of it spent waiting for a chunk + at the barrier, and when the workgroup ran.
`project_profile.py 128 100000 [mask]` - mask adds the measurement-only ablations (1 no stores, 2 no MFMAs)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnnome_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda", 0)
H = int(sys.argv[1]) if len(sys.argv) > 1 else 128
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000
mask = 32 | (int(sys.argv[3]) if len(sys.argv) > 3 else 0)
variant = 20
gen = torch.Generator(device=dev).manual_seed(0)
h = torch.randn(n, H, device=dev, generator=gen)
W = torch.randn(5 * H, H, device=dev, generator=gen) / H ** 0.5
b = torch.randn(5 * H, device=dev, generator=gen)
out = torch.empty(n, 5 * H, device=dev)
planes = ops.weight_planes(W)
tiles = 512   # workgroups of the persistent launch
prof = torch.zeros(tiles * 8, dtype=torch.int64, device=dev)
lib = _lib.load()
ops.set_tuning(2, variant)
ops.set_tuning(1, mask)
for _ in range(3):
    ops.linear(h, W, b, out=out, planes=planes)
lib.gnnome_debug_gate_profile(prof.data_ptr())
s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
ops.linear(h, W, b, out=out, planes=planes)
t.record()
torch.cuda.synchronize()
lib.gnnome_debug_gate_profile(None)
ops.set_tuning(1, 0)
ops.set_tuning(2, 0)
p = prof.view(tiles, 8).cpu().double()
t0 = p[:, 3].min()
start_us, end_us = (p[:, 3] - t0) / 100.0, (p[:, 4] - t0) / 100.0
p = p[p[:, 5] > 0]
units = p[:, 5]
print(f"H={H} N={n} mask={mask}: launch {s.elapsed_time(t) * 1e3:.1f} us, {len(p)} workgroups, {units.mean():.1f} granules each")
life = p[:, 1]
print(f"  life {life.mean():9.0f} cycles = {(life / units).mean():.0f} per granule; row tiles coming in {p[:, 0].mean():.0f} per workgroup; "
      f"waiting for the granule + barrier {(p[:, 2] / units).mean():.0f} per granule")
span = (p[:, 4] - p[:, 3]).mean() / 100.0
print(f"  a workgroup lives {span:.1f} us -> {life.mean() / span / 1e3:.2f} GHz; starts within {start_us.max().item():.1f} us; ends {end_us.min().item():.1f} .. {end_us.max().item():.1f} us")
# where the slow workgroups are: by XCD (block b runs on XCD b % 8) and by position inside the XCD's run
full = prof.view(-1, 8).cpu().double()
idx = torch.arange(full.shape[0])
ok = full[:, 5] > 0
lf = (full[:, 4] - full[:, 3]) / 100.0
print("  life by XCD (us, mean / max): " + "  ".join(f"{x}: {lf[ok & (idx % 8 == x)].mean():.1f}/{lf[ok & (idx % 8 == x)].max():.1f}" for x in range(8)))
order = idx // 8
print("  life by dispatch order inside the XCD (us, mean over XCDs), 8 bins: " + "  ".join(f"{lf[ok & (order >= 8 * k) & (order < 8 * k + 8)].mean():.1f}" for k in range(8)))
print("  granules and row-tile loads of the slowest 5: " + "; ".join(f"wg {int(i)} life {lf[i]:.1f} load {full[i, 0]:.0f}cy wait {full[i, 2]:.0f}cy" for i in torch.argsort(lf, descending=True)[:5]))
print("  ... of the fastest 5: " + "; ".join(f"wg {int(i)} life {lf[i]:.1f} load {full[i, 0]:.0f}cy wait {full[i, 2]:.0f}cy" for i in torch.argsort(torch.where(ok, lf, torch.full_like(lf, 1e9)))[:5]))
