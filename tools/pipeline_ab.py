#!/usr/bin/env python
"""Whole forward with the node projections pipelined under the aggregation's node ranges (engine.aggregate_then_project)
against the one-stream sequence, same process, alternating: ms per forward for engine.PIPELINE_CHUNKS in 1 (off), 2, 4, 8,
eager and replayed from a hipGraph.   usage: pipeline_ab.py [c2|10m|parity64] [chunk list, e.g. 1,2,4,8]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnnome_amd  # noqa: E402
from gnnome_amd import engine, ops  # noqa: E402
from gnnome_amd.capture import CapturedForward  # noqa: E402
from gnnome_amd.synth import make_graph, random_state_dict  # noqa: E402

dev = torch.device("cuda", 0)
wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
n, e, H = {"c2": (100_000, 1_000_000, 128), "10m": (1_000_000, 10_000_000, 128), "parity64": (100_000, 1_000_000, 64)}[wl]
chunks = [int(t) for t in (sys.argv[2] if len(sys.argv) > 2 else "1,2,4,8").split(",")]
reps = 50 if e <= 1_000_000 else 8
g = make_graph(n, e, seed=1)
m = gnnome_amd.SymGatedGCNModel(2, 2, H, 16, 8, 64, "batch").eval()
m.load_state_dict(random_state_dict(H, seed=1))
m.to(dev)
views = ops.GraphViews(g["src"].to(dev), g["dst"].to(dev), n)
x, ef = ops.degree_features(views), g["e"].to(dev)
engine.PIPELINE_MIN_NODES = 0


def timed(fn):
    for _ in range(3):
        out = fn()
    s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        out = fn()
    t.record()
    torch.cuda.synchronize()
    return s.elapsed_time(t) / reps, out


ref = None
with torch.no_grad():
    for rnd in range(2):
        for c in chunks:
            engine.PIPELINE_CHUNKS = c
            ms, out = timed(lambda: m(views, x, ef))
            ref = out.clone() if ref is None else ref
            cap = CapturedForward(m, views, x, ef)
            ms_g, out_g = timed(cap)
            same = torch.equal(out, ref) and torch.equal(out_g, ref)
            print(f"{wl} round {rnd} chunks {c}: eager {ms:.3f} ms, hipGraph {ms_g:.3f} ms per forward, bits equal to the first: {same}", flush=True)
            del cap
