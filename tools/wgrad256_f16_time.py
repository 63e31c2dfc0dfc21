#!/usr/bin/env python
"""Round 6: the 256 x 256 weight-gradient kernel as one-accumulator fp16x3 (ops.wgrad(A, B, amax=...)) against its bf16x6 form (ops.wgrad(A, B)),
alternating in one process: B_3's shape [R,256]^T [R,256] and the projection's [R,1280]^T [R,256]."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnnome_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
for rows, Ka, Kb in ((2_500_000, 256, 256), (250_000, 1280, 256)):
    gen = torch.Generator(device=dev).manual_seed(0)
    A = 1e-4 * torch.randn(rows, Ka, device=dev, generator=gen)
    B = torch.randn(rows, Kb, device=dev, generator=gen)
    amax = A.abs().max().reshape(1).float().view(torch.int32)
    out = torch.empty(Ka, Kb, device=dev)
    want = A[:200_000].double().t() @ B[:200_000].double()
    bound = (A[:200_000].double().abs().t() @ B[:200_000].double().abs()).max().item()
    for rnd in range(3):
        for arm in ("bf16x6", "fp16x3"):
            call = (lambda: ops.wgrad(A, B, out=out, amax=amax)) if arm == "fp16x3" else (lambda: ops.wgrad(A, B, out=out))
            for _ in range(2):
                call()
            evs = []
            for _ in range(10):
                s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                call()
                t.record()
                evs.append((s, t))
            torch.cuda.synchronize()
            ts = sorted(x.elapsed_time(y) for x, y in evs)
            chk = (ops.wgrad(A[:200_000], B[:200_000], amax=amax) if arm == "fp16x3" else ops.wgrad(A[:200_000], B[:200_000])).double()
            print(f"[{rows} x {Ka}]^T [{rows} x {Kb}] round {rnd} {arm}: median {ts[len(ts) // 2]:.4f} ms  min {ts[0]:.4f} ms  "
                  f"max error / max sum|a||b| (first 200k rows) {(chk - want).abs().max().item() / bound:.2e}", flush=True)
