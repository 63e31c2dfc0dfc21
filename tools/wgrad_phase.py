#!/usr/bin/env python
"""In-kernel cycle counters of the 256 x 256 weight-gradient kernel (gnnome_debug_gate_profile): cycles per 16-row slab and the
share of them spent waiting at the slab barrier, per ablation (gnnome_set_tuning key 1)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnnome_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda", 0)
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 2_500_000
gen = torch.Generator(device=dev).manual_seed(0)
A = torch.randn(rows, 256, device=dev, generator=gen)
B = torch.randn(rows, 256, device=dev, generator=gen)
out = torch.empty(256, 256, device=dev)
lib = _lib.load()
for abl in [int(v) for v in (sys.argv[2].split(",") if len(sys.argv) > 2 else "0,5,2,4,1".split(","))]:
    ops.set_tuning(1, abl)
    ops.wgrad(A, B, out=out)
    prof = torch.zeros(256 * 16, dtype=torch.int64, device=dev)
    lib.gnnome_debug_gate_profile(prof.data_ptr())
    ops.wgrad(A, B, out=out)
    torch.cuda.synchronize()
    lib.gnnome_debug_gate_profile(None)
    p = prof.view(256, 4, 4).cpu().double()
    p = p[p[:, 0, 2] > 0]
    per = p[:, :, 0] / p[:, :, 2]
    print(f"ablation {abl}: workgroups {p.shape[0]}  cycles per slab {float(per.mean()):8.0f} (min {float(per.min()):.0f} max {float(per.max()):.0f})  "
          f"at the barrier {float((p[:, :, 1] / p[:, :, 2]).mean()):7.0f}  slabs {float(p[:, :, 2].mean()):.0f}  "
          f"clock {float((p[:, :, 0] / p[:, :, 3].clamp(min=1)).mean()) * 100:.0f} MHz", flush=True)
ops.set_tuning(1, 0)
