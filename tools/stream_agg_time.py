#!/usr/bin/env python
"""Same-process A/B of the two aggregation kernels (gnnome_node_aggregate_f32 against gnnome_node_aggregate_stream_f32), launches
alternating, HIP events around single launches:  python tools/stream_agg_time.py [H] [nodes] [edges] [kind] [chunks,chunks,...]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnnome_amd import ops  # noqa: E402
from gnnome_amd.synth import make_graph  # noqa: E402

dev = torch.device("cuda", 0)
H = int(sys.argv[1]) if len(sys.argv) > 1 else 128
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000
e = int(sys.argv[3]) if len(sys.argv) > 3 else 1_000_000
kind = sys.argv[4] if len(sys.argv) > 4 else "banded"
chunk_list = [int(c) for c in sys.argv[5].split(",")] if len(sys.argv) > 5 else [0]
g = make_graph(n, e, seed=1, kind=kind)
views = ops.GraphViews(g["src"].to(dev), g["dst"].to(dev), n)
gen = torch.Generator(device=dev).manual_seed(0)
ee = torch.randn(e, H, device=dev, generator=gen)
h = torch.randn(n, H, device=dev, generator=gen)
P = torch.randn(n, 5 * H, device=dev, generator=gen)
sc, sh = torch.rand(H, device=dev, generator=gen) * 0.1, torch.randn(H, device=dev, generator=gen)
A1, A2, A3 = (P[:, i * H:(i + 1) * H] for i in range(3))


def timed(fn, reps=30):
    for _ in range(3):
        fn()
    evs = []
    for _ in range(reps):
        s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        t.record()
        evs.append((s, t))
    torch.cuda.synchronize()
    ts = sorted(x.elapsed_time(y) for x, y in evs)
    return ts[len(ts) // 2], ts[0]


def gather():
    ops.STREAM_AGGREGATE = False
    return ops.node_aggregate(ee, A1, A2, A3, views, h, 0, sc, sh)


def stream():
    ops.STREAM_AGGREGATE = "auto"
    return ops.node_aggregate(ee, A1, A2, A3, views, h, 0, sc, sh)


if len(sys.argv) > 6 and sys.argv[6].startswith("once"):   # a few launches of ONE kernel for the PMC passes (tools/pmc_stream.sh)
    fn = stream if sys.argv[6].endswith("stream") else gather
    for _ in range(6):
        fn()
    torch.cuda.synchronize()
    sys.exit(0)
ref = gather()
for chunks in chunk_list:
    views._stream = ops.StreamSchedule(views, chunks=chunks or None)
    s = views._stream
    info = {"H": H, "nodes": n, "edges": e, "kind": kind, "chunks": s.chunks, "far_fraction": round(s.far_fraction, 4), "pending": s.num_pending,
            "overflow": s.num_overflow, "max_live": s.max_live, "max_steps": s.max_steps, "total_steps": s.total_steps, "usable": s.usable, "why": s.why}
    if not s.usable:
        s.usable = True   # measure anyway
    out = stream()
    info["max_abs_diff_vs_gather"] = float((out - ref).abs().max())
    info["scale"] = float(ref.abs().max())
    for rnd in range(3):
        a, b = timed(gather), timed(stream)
        info[f"round{rnd}"] = {"gather_ms_median_min": [round(a[0], 4), round(a[1], 4)], "stream_ms_median_min": [round(b[0], 4), round(b[1], 4)]}
    print(json.dumps(info), flush=True)
