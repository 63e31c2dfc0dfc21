#!/usr/bin/env python
"""Does the aggregation run faster when e' is still on the chip?  (a) per-edge launch time of gnnome_node_aggregate_f32 against the size of e'
(51 MB ... 2 GB at H = 128: the Infinity Cache holds 256 MB); (b) the same launch right after a kernel that WROTE e' (a copy into it: what the gate
leaves behind) against after a kernel that flushed the caches with 1 GB of other traffic.  python tools/agg_scaling.py [H]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnnome_amd import ops  # noqa: E402
from gnnome_amd.synth import make_graph  # noqa: E402

dev = torch.device("cuda", 0)
H = int(sys.argv[1]) if len(sys.argv) > 1 else 128
ops.STREAM_AGGREGATE = False
flush_src = torch.randn(256 * 1024 * 1024, device=dev)   # 1 GB
flush_dst = torch.empty_like(flush_src)
for e in (100_000, 200_000, 400_000, 1_000_000, 2_000_000, 4_000_000):
    n = e // 10
    g = make_graph(n, e, seed=1)
    views = ops.GraphViews(g["src"].to(dev), g["dst"].to(dev), n)
    gen = torch.Generator(device=dev).manual_seed(0)
    ee = torch.randn(e, H, device=dev, generator=gen)
    e2 = torch.randn(e, H, device=dev, generator=gen)
    h = torch.randn(n, H, device=dev, generator=gen)
    P = torch.randn(n, 5 * H, device=dev, generator=gen)
    sc, sh = torch.rand(H, device=dev, generator=gen) * 0.1, torch.randn(H, device=dev, generator=gen)
    A1, A2, A3 = (P[:, i * H:(i + 1) * H] for i in range(3))
    out = torch.empty_like(h)

    def run(prep):
        ts = []
        for _ in range(12):
            prep()
            s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            ops.node_aggregate(ee, A1, A2, A3, views, h, 0, sc, sh, out=out)
            t.record()
            torch.cuda.synchronize()
            ts.append(s.elapsed_time(t))
        ts.sort()
        return ts[len(ts) // 2]

    res = {"H": H, "edges": e, "e_bytes_MB": e * H * 4 / 1e6,
           "back_to_back_ms": run(lambda: None),
           "after_e_was_written_ms": run(lambda: ee.copy_(e2)),
           "after_1GB_of_other_traffic_ms": run(lambda: flush_dst.copy_(flush_src))}
    for k in ("back_to_back_ms", "after_e_was_written_ms", "after_1GB_of_other_traffic_ms"):
        res[k.replace("_ms", "_ns_per_edge")] = round(res[k] * 1e6 / e, 4)
        res[k] = round(res[k], 4)
    print(json.dumps(res), flush=True)
