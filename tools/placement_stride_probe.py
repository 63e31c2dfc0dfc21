#!/usr/bin/env python
"""MEASUREMENT (round 6): is the forward's time a matter of the spacing of the eight XCDs' streams in memory?  Forward time per edge for graph sizes around
configs[1] - among them E = 2^20, where an XCD's eighth of e' is exactly 64 MiB - inside one >= 1 GiB workspace block (placement-independent)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnnome_amd  # noqa: E402
from gnnome_amd import ops  # noqa: E402
from gnnome_amd.graph import views_for  # noqa: E402
from gnnome_amd.synth import make_graph, random_state_dict  # noqa: E402

dev = torch.device("cuda", 0)
hidden = 128
model = gnnome_amd.SymGatedGCNModel(2, 2, hidden, 16, 8, 64, "batch").eval()
model.load_state_dict(random_state_dict(hidden, seed=1))
model.to(dev)
for rnd in range(2):
    for e in (900_000, 983_040, 1_000_000, 1_015_808, 1_048_576, 1_064_960, 1_100_000, 1_179_648):
        n = e // 20 * 2
        g = make_graph(n, e, seed=1)
        views = views_for((g["src"], g["dst"], n), dev)
        x, ef = torch.randn(n, 2, device=dev), g["e"].to(dev)
        model(views, x, ef)
        for k in list(ops._WS_BYTES):
            ops._WS_BYTES[k] = 3 << 30
        for _ in range(5):
            model(views, x, ef)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(60):
            model(views, x, ef)
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 60
        print(f"round {rnd} E = {e:8d} (an eighth of e' = {e * hidden * 4 / 8 / (1 << 20):8.3f} MiB): {ms:.4f} ms = {ms * 1e6 / e:.4f} ns per edge", flush=True)
        ops._WS_BYTES.clear()
