#!/usr/bin/env python
"""MEASUREMENT (round 6): does the forward's time depend on WHERE its buffers lie in HBM?  bench.py runs of one build on one box come out at three levels
(4.13 / 4.22 / 4.26 ms at configs[1]) per PROCESS.  Here one process times the forward over a series of placements: before each phase the previous
workspace-sized block is kept allocated, so the allocator hands the forward fresh memory from the driver."""
import os
import sys
import time

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
held = []


def phase(steps=120):
    for _ in range(10):
        model(views, x, ef)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(steps):
        model(views, x, ef)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / steps


mode = sys.argv[1] if len(sys.argv) > 1 else "hold"
for i in range(int(sys.argv[2]) if len(sys.argv) > 2 else 10):
    ms = phase()
    need = max(ops._WS_BYTES.values())
    probe = torch.empty(need, dtype=torch.uint8, device=dev)      # the block the forward's workspace gets (same size class: the allocator hands it out again)
    addr = probe.data_ptr()
    del probe
    print(f"phase {i} ({mode}): {ms:.4f} ms per forward; workspace {need} B at {addr:#x} (mod 1 GiB {addr % (1 << 30):#x}, mod 2 MiB {addr % (1 << 21):#x}); "
          f"reserved {torch.cuda.memory_reserved() / 1e9:.2f} GB", flush=True)
    if mode == "hold":       # keep a block of the workspace's size class: the next forward's workspace is a new allocation
        held.append(torch.empty(900_000_000 + 4096 * i, dtype=torch.uint8, device=dev))
    elif mode == "same":     # control: nothing changes between phases
        pass
    elif mode == "sleep":    # control: idle time between phases (clocks / temperature)
        time.sleep(2.0)
