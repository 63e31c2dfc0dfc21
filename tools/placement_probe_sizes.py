#!/usr/bin/env python
"""MEASUREMENT (round 6): the forward against the SIZE CLASS of the block its workspace is carved from (the library only asks for >= need bytes): if the
fast placements are those the driver maps with large page-table fragments, a larger (more aligned) allocation should land there more often."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnnome_amd  # noqa: E402
from gnnome_amd import ops  # noqa: E402
from gnnome_amd.graph import views_for  # noqa: E402
from gnnome_amd.synth import make_graph, random_state_dict  # noqa: E402

dev = torch.device("cuda", 0)
n, e, hidden = 100_000, 1_000_000, 128
g = make_graph(n, e, seed=1)
model = gnnome_amd.SymGatedGCNModel(2, 2, hidden, 16, 8, 64, "batch").eval()
model.load_state_dict(random_state_dict(hidden, seed=1))
model.to(dev)
views = views_for((g["src"], g["dst"], n), dev)
x, ef = torch.randn(n, 2, device=dev), g["e"].to(dev)
model(views, x, ef)
key = next(iter(ops._WS_BYTES))
need = ops._WS_BYTES[key]
held = []


def phase(steps=100):
    for _ in range(8):
        model(views, x, ef)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(steps):
        model(views, x, ef)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / steps


GiB = 1 << 30
for size in (need, GiB, GiB + (GiB >> 1), 2 * GiB, 4 * GiB, need):
    times = []
    for i in range(6):
        ops._WS_BYTES[key] = size + (2 << 20) * i       # a size class of its own: a fresh block from the driver
        times.append(phase())
        held.append(torch.empty(size + (2 << 20) * i, dtype=torch.uint8, device=dev))   # takes the block the forward just used out of circulation
    print(f"workspace block of {size / GiB:.3f} GiB: " + "  ".join(f"{t:.4f}" for t in times) + f"   (reserved {torch.cuda.memory_reserved() / 1e9:.1f} GB)", flush=True)
