#!/usr/bin/env python
"""Round 6: the aggregation with a node's pointers and first 20 + 20 neighbour ids in ONE 256-byte record
(gnnome_build_node_records / gnnome_debug_node_records) against the default kernel, alternating in one process; same bits required."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnnome_amd import _lib, ops  # noqa: E402
from gnnome_amd.synth import make_graph  # noqa: E402

dev = torch.device("cuda", 0)
H = int(sys.argv[1]) if len(sys.argv) > 1 else 128
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000
e = 10 * n
g = make_graph(n, e, seed=1)
views = ops.GraphViews(g["src"].to(dev), g["dst"].to(dev), n)
gen = torch.Generator(device=dev).manual_seed(0)
ee = torch.randn(e, H, device=dev, generator=gen)
h = torch.randn(n, H, device=dev, generator=gen)
P = torch.randn(n, 5 * H, device=dev, generator=gen)
sc, sh = torch.rand(H, device=dev, generator=gen) * 0.1, torch.randn(H, device=dev, generator=gen)
A1, A2, A3 = (P[:, i * H:(i + 1) * H] for i in range(3))
lib = _lib.load()
rec = torch.empty(n * 64, dtype=torch.int32, device=dev)
_lib.check(lib.gnnome_build_node_records(*(ctypes.c_void_p(t.data_ptr()) for t in (views.in_ptr, views.srt_src, views.out_ptr, views.out_pos, views.out_dst)), n,
                                         ctypes.c_void_p(rec.data_ptr()), None), "build_node_records")
torch.cuda.synchronize()
deg_in, deg_out = views.in_ptr[1:] - views.in_ptr[:-1], views.out_ptr[1:] - views.out_ptr[:-1]
print(f"H={H} N={n}: {float(((deg_in <= 20) & (deg_out <= 20)).float().mean()):.1%} of the nodes fit their record")
ref = None
for rnd in range(3):
    for arm in ("default", "records"):
        lib.gnnome_debug_node_records(ctypes.c_void_p(rec.data_ptr()) if arm == "records" else None)
        for _ in range(3):
            out = ops.node_aggregate(ee, A1, A2, A3, views, h, 0, sc, sh)
        evs = []
        for _ in range(30):
            s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            out = ops.node_aggregate(ee, A1, A2, A3, views, h, 0, sc, sh)
            t.record()
            evs.append((s, t))
        torch.cuda.synchronize()
        ts = sorted(x.elapsed_time(y) for x, y in evs)
        if ref is None:
            ref = out.clone()
        print(f"round {rnd} {arm:8s}: median {ts[len(ts) // 2]:.4f} ms  min {ts[0]:.4f} ms  same bits as the default {bool(torch.equal(out, ref))}", flush=True)
lib.gnnome_debug_node_records(None)
