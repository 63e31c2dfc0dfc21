#!/usr/bin/env python
"""Launch time of the weight-gradient product C[Ka,Kb] = A[R,Ka]^T B[R,Kb] (gnnome_wgrad_f32) with the kernel's measurement-only
ablations (gnnome_set_tuning key 1: 1 = no re-fetch of slabs, 2 = no MFMAs, 4 = no staging, 5 = neither loads nor staging, 8 = two slabs ahead, 16 = fp16x3 without a scale)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnnome_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 2_500_000
Ka = int(sys.argv[2]) if len(sys.argv) > 2 else 256
Kb = int(sys.argv[3]) if len(sys.argv) > 3 else 256
abls = [int(v) for v in (sys.argv[4].split(",") if len(sys.argv) > 4 else "0,1,2,4,5".split(","))]
gen = torch.Generator(device=dev).manual_seed(0)
A = torch.randn(rows, Ka, device=dev, generator=gen)
B = torch.randn(rows, Kb, device=dev, generator=gen)
out = torch.empty(Ka, Kb, device=dev)
flops = 2.0 * rows * Ka * Kb * 6
want = A.double().t() @ B.double()
bound = (A.double().abs().t() @ B.double().abs()).max().item()
for rnd in range(2):
    for abl in abls:
        ops.set_tuning(1, abl)
        for _ in range(2):
            ops.wgrad(A, B, out=out)
        evs = []
        for _ in range(10):
            s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            ops.wgrad(A, B, out=out)
            t.record()
            evs.append((s, t))
        torch.cuda.synchronize()
        ts = sorted(x.elapsed_time(y) for x, y in evs)
        med = ts[len(ts) // 2]
        err = (out.double() - want).abs().max().item() / bound
        print(f"round {rnd} ablation {abl}: median {med:.4f} ms  {flops / med / 1e9:.0f} bf16 TF/s  {(rows * (Ka + Kb) * 4) / med / 1e9:.2f} TB/s  "
              f"max error / max sum|a||b| {err:.2e}", flush=True)
ops.set_tuning(1, 0)
