#!/usr/bin/env python
"""Launch time of gnnome_agg_edge_bwd_stats_f32 (the largest kernel of the training step) at configs[2]'s size: all nine row loads
in flight together (default) against the compiler's own placement (gnnome_set_tuning(4, 77): every load sunk to its first use)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnnome_amd import ops  # noqa: E402
from gnnome_amd.synth import make_graph  # noqa: E402

dev = torch.device("cuda", 0)
n, e, H = 100_000, 1_000_000, 128
g = make_graph(n, e, seed=1)
views = ops.GraphViews(g["src"].to(dev), g["dst"].to(dev), n)
gen = torch.Generator(device=dev).manual_seed(0)
r = lambda *s: torch.randn(*s, device=dev, generator=gen)  # noqa: E731
ee, xe, de = r(e, H), r(e, H), r(e, H)
Tf, Uf, Tb, Ub, P = r(n, H), r(n, H), r(n, H), r(n, H), r(n, 2 * H)
scale, shift, mean = r(H), r(H), r(H)
ref = None
for rnd in range(3):
    for knob in (0, 77):
        ops.set_tuning(4, knob)
        d0 = de.clone()
        for _ in range(3):
            ops.agg_edge_bwd_stats(ee, Tf, Uf, Tb, Ub, P[:, :H], P[:, H:], views, d0, xe, scale, shift, mean)
        evs = []
        for _ in range(20):
            s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            ops.agg_edge_bwd_stats(ee, Tf, Uf, Tb, Ub, P[:, :H], P[:, H:], views, d0, xe, scale, shift, mean)
            t.record()
            evs.append((s, t))
        torch.cuda.synchronize()
        ts = sorted(a.elapsed_time(b) for a, b in evs)
        d1 = de.clone()
        _, s1, s2 = ops.agg_edge_bwd_stats(ee, Tf, Uf, Tb, Ub, P[:, :H], P[:, H:], views, d1, xe, scale, shift, mean)
        ref = (d1, s1.clone(), s2.clone()) if ref is None else ref
        same = torch.equal(d1, ref[0]) and torch.equal(s1, ref[1]) and torch.equal(s2, ref[2])
        print(f"round {rnd} {'compiler placement' if knob else 'loads batched     '}: median {ts[len(ts) // 2]:.4f} ms  min {ts[0]:.4f} ms  bits equal: {same}", flush=True)
ops.set_tuning(4, 0)
