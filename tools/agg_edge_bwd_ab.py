#!/usr/bin/env python
"""Same-process A/B of gnnome_agg_edge_bwd_stats_f32's forms (tuning 4: 77 = the three streams as plain loads, 0 = the default: nontemporal) and of the
fused launch (gnnome_agg_bwd_fused_f32) against the pair it replaces, launches alternating, HIP events around single calls, results compared:
python tools/agg_edge_bwd_ab.py [H] [nodes] [edges] [kind] [membw]"""
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
g = make_graph(n, e, seed=1, kind=kind)
views = ops.GraphViews(g["src"].to(dev), g["dst"].to(dev), n)
gen = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s: torch.randn(*s, device=dev, generator=gen)   # noqa: E731
ee, xe, de0 = rnd(e, H), rnd(e, H), rnd(e, H)
Tf, Uf, Tb, Ub = rnd(n, H), rnd(n, H), rnd(n, H), rnd(n, H)
P = rnd(n, 5 * H)
A2, A3 = P[:, H:2 * H], P[:, 2 * H:3 * H]
sc, sh, mn = torch.rand(H, device=dev, generator=gen) + 0.5, rnd(H), rnd(H)
KEY = 4   # kTuneGateExperiment (csrc/common.h)


def run(variant, de):
    ops.set_tuning(KEY, variant)
    try:
        return ops.agg_edge_bwd_stats(ee, Tf, Uf, Tb, Ub, A2, A3, views, de, xe, sc, sh, mn)
    finally:
        ops.set_tuning(KEY, 0)


VARIANTS = (77, 0)
outs = {}
for v in VARIANTS:
    de = de0.clone()
    _, s1, s2 = run(v, de)
    outs[v] = (de, s1.clone(), s2.clone())
torch.cuda.synchronize()
rel = lambda a, b: ((a - b).abs().max() / b.abs().max()).item()   # noqa: E731
for v in VARIANTS[1:]:
    print(json.dumps({"variant": v, "de_bit_equal_to_77": torch.equal(outs[77][0], outs[v][0]), "s1_rel": rel(outs[v][1], outs[77][1]),
                      "s2_rel": rel(outs[v][2], outs[77][2])}))

scratch = de0.clone()
times = {v: [] for v in VARIANTS}
for rep in range(40):
    for v in VARIANTS:
        s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        run(v, scratch)
        t.record()
        times[v].append((s, t))
torch.cuda.synchronize()
rec = {"H": H, "nodes": n, "edges": e, "kind": kind}
for v, name in ((77, "plain_loads_ms"), (0, "default_nontemporal_ms")):
    ts = sorted(a.elapsed_time(b) for a, b in times[v][5:])
    rec[name] = round(ts[len(ts) // 2], 4)
    rec[name + "_min"] = round(ts[0], 4)
print(json.dumps(rec))

# ---- the pair (node sums, then the per-edge pass) against the one fused launch
def pair(de):
    si, so = ops.node_aggregate_raw(ee, None, Tb, Tf, views, 2, n)
    _, s1, s2 = ops.agg_edge_bwd_stats(ee, Tf, Uf, Tb, Ub, A2, A3, views, de, xe, sc, sh, mn)
    return si, so, de, s1, s2


def fused(de):
    return ops.agg_bwd_fused(ee, Tf, Uf, Tb, Ub, A2, A3, views, de, xe, sc, sh, mn, n)


def fused_variant(v):
    def run_(de):
        ops.set_tuning(KEY, v)
        try:
            return fused(de)
        finally:
            ops.set_tuning(KEY, 0)
    return run_


a, b = pair(de0.clone()), fused(de0.clone())
torch.cuda.synchronize()
print(json.dumps({"fused_vs_pair": {k: rel(y.clone(), x.clone()) for k, x, y in zip(("sum_in", "sum_out", "de", "s1", "s2"), a, b)},
                  "fused_same_bits_twice": all(torch.equal(u, v) for u, v in zip(fused(de0.clone()), b))}))
cases = (("pair_ms", pair), ("fused_ms", fused), ("fused_4_waves_2_4_items_ms", fused_variant(88)))
times = {name: [] for name, _ in cases}
for rep in range(40):
    for name, fn in cases:
        s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fn(scratch)
        t.record()
        times[name].append((s, t))
torch.cuda.synchronize()
rec = {"H": H, "nodes": n, "edges": e, "kind": kind}
for name in times:
    ts = sorted(x.elapsed_time(y) for x, y in times[name][5:])
    rec[name] = round(ts[len(ts) // 2], 4)
    rec[name + "_min"] = round(ts[0], 4)
print(json.dumps(rec))

if len(sys.argv) > 5 and sys.argv[5] == "membw":   # what the chip gives plain streams of the same 512 MB tensors (reads cap lower than read + write)
    def med(fn, reps=20):
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
        ts = sorted(a.elapsed_time(b) for a, b in evs)
        return ts[len(ts) // 2]
    out = torch.empty_like(ee)
    gb = ee.numel() * 4 / 1e9
    bw = {}
    for name, fn, traffic in (("read_colsum2", lambda: ops.colsum2(ee), gb), ("read_torch_sum", lambda: ee.sum(), gb),
                              ("read2_colsum2_xy", lambda: ops.colsum2(ee, xe), 2 * gb),
                              ("copy", lambda: out.copy_(ee), 2 * gb), ("inplace_mul", lambda: out.mul_(1.0001), 2 * gb),
                              ("add3", lambda: torch.add(ee, xe, out=out), 3 * gb), ("fill", lambda: out.fill_(1.0), gb)):
        ms = med(fn)
        bw[name] = {"ms": round(ms, 4), "TB_per_s": round(traffic / ms, 3)}
    print(json.dumps({"plain_streams": bw}))
