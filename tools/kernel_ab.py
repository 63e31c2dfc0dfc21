#!/usr/bin/env python
"""Within-process A/B timing of the layer kernels at a BASELINE workload size (GPU box only).

Times each kernel (i) back-to-back with itself and (ii) in the real layer sequence, with HIP events on
the launch stream, so that cache-state / clock interactions between neighbouring kernels are visible.
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnnome_amd import ops  # noqa: E402
from gnnome_amd.synth import make_graph  # noqa: E402


def timed(fn, reps):
    evs = []
    for _ in range(reps):
        s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        t.record()
        evs.append((s, t))
    torch.cuda.synchronize()
    ts = sorted(s.elapsed_time(t) for s, t in evs)
    return ts[len(ts) // 2], ts[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nodes", type=int, default=100_000)
    ap.add_argument("--edges", type=int, default=1_000_000)
    ap.add_argument("--hidden", type=int, default=128)
    ap.add_argument("--reps", type=int, default=20)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    n, e, H = a.nodes, a.edges, a.hidden
    g = make_graph(n, e, seed=1)
    views = ops.GraphViews(g["src"].to(dev), g["dst"].to(dev), n)
    gen = torch.Generator(device=dev).manual_seed(0)
    ee = torch.randn(e, H, device=dev, generator=gen)
    h = torch.randn(n, H, device=dev, generator=gen)
    P = torch.randn(n, 5 * H, device=dev, generator=gen)
    Wcat = torch.randn(5 * H, H, device=dev, generator=gen) / H ** 0.5
    bcat = torch.randn(5 * H, device=dev, generator=gen)
    W3 = torch.randn(H, H, device=dev, generator=gen) / H ** 0.5
    sc, sh = torch.rand(H, device=dev, generator=gen) * 0.1, torch.randn(H, device=dev, generator=gen)
    A1, A2, A3, B1, B2 = (P[:, i * H:(i + 1) * H] for i in range(5))

    gate = lambda: ops.edge_gate(ee, B1, B2, views, W3, 0, sc, sh)  # noqa: E731
    agg = lambda: ops.node_aggregate(ee, A1, A2, A3, views, h, 0, sc, sh)  # noqa: E731
    lin = lambda: ops.linear(h, Wcat, bcat, out=P)  # noqa: E731
    for _ in range(3):
        lin(), gate(), agg()
    torch.cuda.synchronize()
    print(f"N={n} E={e} H={H}   median / min ms")
    for name, fn in (("linear", lin), ("edge_gate", gate), ("node_aggregate", agg)):
        med, mn = timed(fn, a.reps)
        print(f"  isolated  {name:16s} {med:.3f} / {mn:.3f}")
    seq = {"linear": [], "edge_gate": [], "node_aggregate": []}
    for _ in range(a.reps):
        for name, fn in (("linear", lin), ("edge_gate", gate), ("node_aggregate", agg)):
            s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            fn()
            t.record()
            seq[name].append((s, t))
    torch.cuda.synchronize()
    for name, evs in seq.items():
        ts = sorted(s.elapsed_time(t) for s, t in evs)
        print(f"  in-layer  {name:16s} {ts[len(ts) // 2]:.3f} / {ts[0]:.3f}")
    # edge-gate variants, interleaved rounds (cdna_hip_programming.md 5.4 rule 24)
    names = {1: "tile-per-workgroup, exact-fp32 MFMA", 6: "wave-specialised, exact-fp32 MFMA, barrier hand-over",
             5: "wave-specialised, exact-fp32 MFMA, counter hand-over", 0: "bf16x6 edge-tile kernel (default)"}
    res = {k: [] for k in names}
    for _ in range(5):
        for k in names:
            ops.set_tuning(0, k)
            res[k].append(timed(gate, 5)[0])
    ops.set_tuning(0, 0)
    for k, v in res.items():
        print(f"  edge_gate variant {names[k]:44s} median {sorted(v)[len(v) // 2]:.3f}  min {min(v):.3f} ms")
    ops.set_tuning(0, 5)
    for order, nm in ((1, "chunked"), (0, "interleaved, XCD-contiguous")):
        ops.set_tuning(3, order)
        print(f"  wave-specialised gate, tile order {nm:28s} median {timed(gate, 10)[0]:.3f} ms")
    ops.set_tuning(3, 0)
    for abl_, nm in ((0, "plain loads/stores"), (16, "nontemporal e loads + e' stores")):
        ops.set_tuning(1, abl_)
        print(f"  wave-specialised gate, {nm:36s} median {timed(gate, 10)[0]:.3f} ms")
    ops.set_tuning(1, 0)
    ops.set_tuning(0, 0)
    lv = {1: "tile kernel, bf16x6", 2: "weight-stationary, exact-fp32 MFMA", 3: "weight-stationary, bf16x6, A staged in LDS",
          0: "streaming, bf16x6, barrier-free (default)"}
    res = {k: [] for k in lv}
    for _ in range(5):
        for k in lv:
            ops.set_tuning(2, k)
            res[k].append(timed(lin, 5)[0])
    ops.set_tuning(2, 0)
    for k, v in res.items():
        print(f"  linear [N,{H}]x[{H},{5 * H}] variant {lv[k]:44s} median {sorted(v)[len(v) // 2]:.3f}  min {min(v):.3f} ms")
    # ablations (results are wrong by construction; timing only)
    abl = {0: "full", 1: "no node gathers", 2: "no e_out stores", 4: "no HBM tile loads", 7: "MFMA + LDS only"}
    res = {k: [] for k in abl}
    for _ in range(3):
        for k in abl:
            ops.set_tuning(1, k)
            res[k].append(timed(gate, 5)[0])
    ops.set_tuning(1, 0)
    for k, v in res.items():
        print(f"  bf16x6 gate ablation {abl[k]:24s} median {sorted(v)[len(v) // 2]:.3f} ms")
    abl = {0: "full", 1: "no node gathers", 2: "no e_out stores", 4: "no HBM tile loads", 8: "no MFMA", 7: "MFMA + LDS only",
           15: "loop skeleton only"}
    ops.set_tuning(0, 5)
    res = {k: [] for k in abl}
    for _ in range(3):
        for k in abl:
            ops.set_tuning(1, k)
            res[k].append(timed(gate, 5)[0])
    ops.set_tuning(1, 0)
    ops.set_tuning(0, 0)
    for k, v in res.items():
        print(f"  exact-fp32 wave-specialised gate ablation {abl[k]:24s} median {sorted(v)[len(v) // 2]:.3f} ms")
    # raw copy bandwidth reference on the same tensors
    dst = torch.empty_like(ee)
    med, mn = timed(lambda: dst.copy_(ee), a.reps)
    print(f"  torch copy of e ({ee.numel() * 4 / 1e6:.0f} MB r + w): {med:.3f} ms -> {2 * ee.numel() * 4 / med / 1e9:.2f} TB/s")


if __name__ == "__main__":
    main()
