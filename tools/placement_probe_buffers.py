#!/usr/bin/env python
"""MEASUREMENT (round 6): WHICH of the forward's buffers decides the level?  The forward takes h0, h1, P, e, PQ as allocations of their own
(gnnome_model_forward_buffers_f32); between phases one size class is taken out of circulation, so that only the buffer(s) of that size move to fresh memory."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnnome_amd  # noqa: E402
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


def where():
    """Addresses of the blocks the forward's next e and P will get (the allocator hands a freed block of the same size class out again)."""
    te, tp = torch.empty((e, hidden), dtype=torch.float32, device=dev), torch.empty((n, 5 * hidden), dtype=torch.float32, device=dev)
    a = (te.data_ptr(), tp.data_ptr())
    del te, tp
    return a


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


sizes = {"e [E,H]": (e, hidden, 1), "P [N,5H]": (n, 5 * hidden, 1), "h0, h1, PQ [N,H]": (n, hidden, 3)}
print(f"start: {phase():.4f} ms", flush=True)
for rnd in range(2):
    for name, (rows, cols, count) in sizes.items():
        times, notes = [], []
        for i in range(6):
            for _ in range(count):
                held.append(torch.empty((rows, cols), dtype=torch.float32, device=dev))   # takes the cached block(s) of this size: the forward's next one is fresh
            times.append(phase())
            ae, ap = where()
            notes.append(f"{times[-1]:.4f} ms  e at {ae:#x}  P at {ap:#x}  (P mod 1 GiB = {(ap % (1 << 30)) / (1 << 20):7.1f} MiB, mod 512 MiB = {(ap % (1 << 29)) / (1 << 20):6.1f}, "
                         f"P - e mod 512 MiB = {((ap - ae) % (1 << 29)) / (1 << 20):6.1f})")
        print(f"round {rnd} fresh {name:18s}: " + "  ".join(f"{t:.4f}" for t in times) + f"   (reserved {torch.cuda.memory_reserved() / 1e9:.1f} GB)", flush=True)
        for ln in notes:
            print("      " + ln, flush=True)
