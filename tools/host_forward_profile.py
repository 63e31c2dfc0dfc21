#!/usr/bin/env python
"""Host cost of one eval forward (model(graph, x, e) -> logits) on a graph so small that the GPU is never the bottleneck: wall time per call with
the one-call entry (gnnome_model_forward_f32) and call by call, and cProfile's top entries for the one-call path."""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnnome_amd  # noqa: E402
from gnnome_amd import engine  # noqa: E402
from gnnome_amd.graph import views_for  # noqa: E402
from gnnome_amd.synth import make_graph, random_state_dict  # noqa: E402

dev = torch.device("cuda", 0)
n, e, hidden = 2000, 20000, int(sys.argv[1]) if len(sys.argv) > 1 else 64
g = make_graph(n, e, seed=1)
model = gnnome_amd.SymGatedGCNModel(2, 2, hidden, 16, 8, 64, "batch").eval()
model.load_state_dict(random_state_dict(hidden, seed=1))
model.to(dev)
views = views_for((g["src"], g["dst"], n), dev)
x, ef = torch.randn(n, 2, device=dev), g["e"].to(dev)


def run(k):
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(k):
        out = model(views, x, ef)
    host = time.perf_counter() - t
    torch.cuda.synchronize()
    return host / k * 1e6, (time.perf_counter() - t) / k * 1e6


for one in (True, False, True):
    engine.ONE_CALL_FORWARD = one
    run(20)
    h, w = run(500)
    print(f"one_call={one}: host {h:.0f} us per forward, wall {w:.0f} us per forward (N={n} E={e} H={hidden})")
engine.ONE_CALL_FORWARD = True
pr = cProfile.Profile()
pr.enable()
run(500)
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(18)
