#!/usr/bin/env python
"""Upper bounds on the two byte-saving fusions of the training step that are NOT built, measured before
building them - what each could save per layer at configs[2] if its work came for free inside the kernel that would absorb it:

  (a) dB2h inside mode 3 of the edge-tile kernel: gnnome_segment_sum2_f32 (both transposes of the gate's gathers in one launch) against the
      out-edge half alone (gnnome_segment_sum_f32 through out_pos) - the in-edge half is what mode 3 would take over;
  (b) the in-edge half of the forward aggregation inside the gate's store pass: the aggregation with that half switched off
      (gnnome_set_tuning(7, 7); wrong results, time only) against the whole kernel.

    python tools/train_fusion_bounds.py [hidden]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnnome_amd import ops  # noqa: E402
from gnnome_amd.synth import make_graph  # noqa: E402

dev = torch.device("cuda", 0)
H = int(sys.argv[1]) if len(sys.argv) > 1 else 128
n, e = 100_000, 1_000_000
g = make_graph(n, e, seed=1)
views = ops.GraphViews(g["src"].to(dev), g["dst"].to(dev), n)
gen = torch.Generator(device=dev).manual_seed(0)
dxe = torch.randn(e, H, device=dev, generator=gen)
ee = torch.randn(e, H, device=dev, generator=gen)
h = torch.randn(n, H, device=dev, generator=gen)
P = torch.randn(n, 5 * H, device=dev, generator=gen)
sc, sh = torch.rand(H, device=dev, generator=gen) * 0.1, torch.randn(H, device=dev, generator=gen)
A1, A2, A3 = (P[:, i * H:(i + 1) * H] for i in range(3))
flush = torch.empty(1 << 28, dtype=torch.uint8, device=dev)   # 256 MiB = the MALL: every timed launch starts from HBM like inside the step


def timed(fn, reps=30):
    for _ in range(3):
        fn()
    evs = []
    for _ in range(reps):
        flush.zero_()
        s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        t.record()
        evs.append((s, t))
    torch.cuda.synchronize()
    ts = sorted(x.elapsed_time(y) for x, y in evs)
    return ts[len(ts) // 2], ts[0]


for rnd in range(3):
    both = timed(lambda: ops.segment_sum2(dxe, views, n))
    out_only = timed(lambda: ops.segment_sum(dxe, views.out_ptr, views.out_pos, n))
    in_only = timed(lambda: ops.segment_sum(dxe, views.in_ptr, None, n))
    print(f"round {rnd} H={H} (a) segment_sum2 {both[0]:.4f} ms | out-edge half alone {out_only[0]:.4f} | in-edge half alone {in_only[0]:.4f} | "
          f"bound on dB2h inside mode 3: {both[0] - out_only[0]:+.4f} ms per layer", flush=True)
    ops.set_tuning(7, 0)
    whole = timed(lambda: ops.node_aggregate(ee, A1, A2, A3, views, h, 0, sc, sh))
    ops.set_tuning(7, 7)
    half = timed(lambda: ops.node_aggregate(ee, A1, A2, A3, views, h, 0, sc, sh))
    ops.set_tuning(7, 0)
    print(f"round {rnd} H={H} (b) aggregation {whole[0]:.4f} ms | without its in-edge half {half[0]:.4f} | "
          f"bound on the in-edge half inside the gate: {whole[0] - half[0]:+.4f} ms per layer", flush=True)
