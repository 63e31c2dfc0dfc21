#!/usr/bin/env python
"""MEASUREMENT: is there anything to win by running a layer's gate and an aggregation at the same time?

Within a layer the two are a chain (the aggregation reads the gate's output); only a split of the graph into node ranges would let the aggregation
of range c run beside the gate of range c + 1.  Before building that, this script measures the bound: the two kernels on INDEPENDENT buffers,
(a) one after the other on one stream, (b) on two streams at once, whole-size and as halves (gate of one half of the edges beside the aggregation of
one half of the nodes, twice).  Results are compared bit for bit with the sequential launches.
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnnome_amd import ops  # noqa: E402
from gnnome_amd.synth import make_graph  # noqa: E402

dev = torch.device("cuda", 0)
H = int(sys.argv[1]) if len(sys.argv) > 1 else 128
n, e = (100_000, 1_000_000) if H <= 128 else (250_000, 2_500_000)
g = make_graph(n, e, seed=1)
views = ops.GraphViews(g["src"].to(dev), g["dst"].to(dev), n)
gen = torch.Generator(device=dev).manual_seed(0)
e_in = torch.randn(e, H, device=dev, generator=gen)
e_out = torch.empty_like(e_in)
e_agg = torch.randn(e, H, device=dev, generator=gen)
h = torch.randn(n, H, device=dev, generator=gen)
h_out = torch.empty_like(h)
P = torch.randn(n, 5 * H, device=dev, generator=gen)
W3 = torch.randn(H, H, device=dev, generator=gen) / H ** 0.5
sc, sh = torch.rand(H, device=dev, generator=gen) * 0.1, torch.randn(H, device=dev, generator=gen)
A1, A2, A3, B1, B2 = (P[:, i * H:(i + 1) * H] for i in range(5))
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
half_n = n // 2
half_e = int(views.in_ptr[half_n].item()) if hasattr(views, "in_ptr") else e // 2


def gate(num_edges=None):
    ops.edge_gate(e_in, B1, B2, views, W3, 0, sc, sh, out=e_out, num_edges=num_edges)


def agg(node_range=None):
    ops.node_aggregate(e_agg, A1, A2, A3, views, h, 0, sc, sh, node_range=node_range, out=h_out)


def timed(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2], ts[0]


def sequential():
    gate()
    agg()


def concurrent():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur)
    s2.wait_stream(cur)
    with torch.cuda.stream(s1):
        gate()
    with torch.cuda.stream(s2):
        agg()
    cur.wait_stream(s1)
    cur.wait_stream(s2)


def concurrent_agg_first():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur)
    s2.wait_stream(cur)
    with torch.cuda.stream(s2):
        agg()
    with torch.cuda.stream(s1):
        gate()
    cur.wait_stream(s1)
    cur.wait_stream(s2)


sequential()
torch.cuda.synchronize()
ref_e, ref_h = e_out.clone(), h_out.clone()
for rnd in range(3):
    for name, fn in (("gate alone", gate), ("aggregation alone", agg), ("one after the other", sequential), ("two streams (gate first)", concurrent),
                     ("two streams (aggregation first)", concurrent_agg_first)):
        e_out.zero_()
        h_out.zero_()
        med, mn = timed(fn)
        same = (torch.equal(e_out, ref_e) if "aggregation alone" != name else True) and (torch.equal(h_out, ref_h) if "gate alone" != name else True)
        print(f"H={H} round {rnd} {name:34s}: median {med:.4f} ms  min {mn:.4f} ms  same bits {same}", flush=True)
