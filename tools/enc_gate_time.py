#!/usr/bin/env python
"""Launch time of the layer-0 gate with the folded edge encoder (gnnome_edge_gate_encode_f32) at configs[1]'s size, per gate
variant (gnnome_set_tuning key 0): 0 = k_edge_gate_enc16, 7 = plane form with an in-register encoder, 8 = second generation."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnnome_amd import ops
from gnnome_amd.synth import make_graph
dev = torch.device("cuda", 0)
e, H = 1_000_000, 128; n = e // 10
g = make_graph(n, e, seed=1)
views = ops.GraphViews(g["src"].to(dev), g["dst"].to(dev), n)
gen = torch.Generator(device=dev).manual_seed(0)
P = torch.randn(n, 5 * H, device=dev, generator=gen)
W3 = torch.randn(H, H, device=dev, generator=gen) / H ** 0.5
sc, sh = torch.rand(H, device=dev, generator=gen) * 0.1, torch.randn(H, device=dev, generator=gen)
enc = (torch.randn(16, 2, device=dev, generator=gen), torch.randn(16, device=dev, generator=gen), torch.randn(H, 16, device=dev, generator=gen) / 4,
       torch.randn(H, device=dev, generator=gen))
e_raw = g["e"].to(dev)
for rnd in range(2):
    for v in (0, 7, 8):
        ops.set_tuning(0, v)
        for _ in range(5):
            ops.edge_gate_encode(e_raw, enc, P[:, 3 * H:4 * H], P[:, 4 * H:], views, W3, sc, sh)
        s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(50):
            ops.edge_gate_encode(e_raw, enc, P[:, 3 * H:4 * H], P[:, 4 * H:], views, W3, sc, sh)
        t.record(); torch.cuda.synchronize()
        print(f"round {rnd} variant {v}: {s.elapsed_time(t) / 50:.4f} ms (incl. the fold kernel and the output allocation)", flush=True)
ops.set_tuning(0, 0)
