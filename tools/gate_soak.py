#!/usr/bin/env python
"""Soak test of the counter-synchronised edge-tile kernels: many launches at random sizes (ragged last tiles, fewer tiles
than workgroups, one edge), every mode, each checked against the barrier-synchronised / tile kernels behind the same entry
points.  A lost hand-over would trap (or hang until the caller's timeout); a race shows up as a mismatch."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnnome_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
gen = torch.Generator().manual_seed(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
launches = int(sys.argv[2]) if len(sys.argv) > 2 else 1200
t0 = time.time()
worst = 0.0
for it in range(launches):
    H = (64, 128)[int(torch.randint(0, 2, (1,), generator=gen))]
    kind = int(torch.randint(0, 10, (1,), generator=gen))
    E = int(torch.randint(1, 40, (1,), generator=gen)) if kind == 0 else int(torch.randint(1, 9000, (1,), generator=gen)) if kind < 4 \
        else int(torch.randint(9000, 400_000, (1,), generator=gen))
    n = max(2, E // 7)
    src = torch.randint(0, n, (E,), generator=gen, dtype=torch.int64).int()
    dst = torch.randint(0, n, (E,), generator=gen, dtype=torch.int64).int()
    views = ops.GraphViews(src.to(dev), dst.to(dev), n)
    e = (3.0 * torch.randn(E, H, generator=gen)).to(dev)
    P = torch.randn(n, 5 * H, generator=gen).to(dev)
    W3 = (torch.randn(H, H, generator=gen) / H ** 0.5).to(dev)
    sc, sh = (0.5 + torch.rand(H, generator=gen)).to(dev), torch.randn(H, generator=gen).to(dev)
    B1, B2 = P[:, 3 * H:4 * H], P[:, 4 * H:]
    mode = it % 3
    if mode == 0:    # the gate, in place and out of place
        got = ops.edge_gate(e.clone(), B1, B2, views, W3, 0, sc, sh)
        ops.set_tuning(0, 6)
        want = ops.edge_gate(e.clone(), B1, B2, views, W3, 0, sc, sh)
        ops.set_tuning(0, 0)
    elif mode == 1:  # raw gate + statistics
        got, mean, var = ops.edge_gate_raw_stats(e, B1, B2, views, W3)
        want = ops.edge_gate_raw(e, B1, B2, views, W3)
        m2, v2 = ops.batch_stats(want)
        assert (mean - m2).abs().max().item() < 1e-3 * max(1.0, m2.abs().max().item()), (it, E, H, "mean")
    else:            # C += A W^T (only takes the edge-tile kernel from 32768 rows)
        C = torch.randn(E, H, generator=gen).to(dev)
        want = C + e @ W3.t()
        got = ops.linear(e, W3, None, out=C, accumulate=True)
    err = (got - want).abs().max().item() / max(1.0, want.abs().max().item())
    worst = max(worst, err)
    assert err < 2e-5, (it, E, H, mode, err)
    if it % 200 == 199:
        torch.cuda.synchronize()
        print(f"{it + 1} launches ok, worst relative deviation {worst:.2e}, {time.time() - t0:.0f} s", flush=True)
torch.cuda.synchronize()
print(f"soak ok: {launches} launches, worst relative deviation {worst:.2e}")
