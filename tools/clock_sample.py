#!/usr/bin/env python
"""Shader clock and power while one kernel runs back to back (the phase counters of round 4 put the shader clock at 1.22-1.24 GHz
under the H = 256 edge-tile kernel against 2.0 under the H = 128 one - is that kernel power-capped?).  rocm-smi is polled from a thread while the main
thread keeps the queue full of launches of ONE kernel.  python tools/clock_sample.py"""
import json
import os
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnnome_amd import ops  # noqa: E402
from gnnome_amd.synth import make_graph  # noqa: E402

dev = torch.device("cuda", 0)


def poll(stop, out):
    while not stop.is_set():
        try:
            r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=5)
            d = json.loads(r.stdout)
            card = d[sorted(k for k in d if k.startswith("card"))[0]]
            out.append({k: v for k, v in card.items() if "sclk" in k.lower() or "power" in k.lower() or "mclk" in k.lower()})
        except Exception as ex:  # noqa: BLE001
            out.append({"error": str(ex)[:100]})
        time.sleep(0.15)


def run(name, fn, seconds=4.0):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    stop, samples = threading.Event(), []
    th = threading.Thread(target=poll, args=(stop, samples))
    th.start()
    t0, launches = time.perf_counter(), 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(50):
            fn()
        launches += 50
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    stop.set()
    th.join()
    print(json.dumps({"kernel": name, "ms_per_launch_wall": dt / launches * 1e3, "samples": samples[1:-1][:12]}), flush=True)


idle = []
st = threading.Event()
th = threading.Thread(target=poll, args=(st, idle))
th.start()
time.sleep(1.0)
st.set()
th.join()
print(json.dumps({"kernel": "idle", "samples": idle[:3]}), flush=True)
for H, e in ((128, 1_000_000), (256, 2_500_000)):
    n = e // 10
    g = make_graph(n, e, seed=1)
    views = ops.GraphViews(g["src"].to(dev), g["dst"].to(dev), n)
    gen = torch.Generator(device=dev).manual_seed(0)
    ee = torch.randn(e, H, device=dev, generator=gen)
    out = torch.empty_like(ee)
    P = torch.randn(n, 5 * H, device=dev, generator=gen)
    h = torch.randn(n, H, device=dev, generator=gen)
    W3 = torch.randn(H, H, device=dev, generator=gen) / H ** 0.5
    sc, sh = torch.rand(H, device=dev, generator=gen) * 0.1, torch.randn(H, device=dev, generator=gen)
    run(f"edge gate H={H} E={e}", lambda: ops.edge_gate(ee, P[:, 3 * H:4 * H], P[:, 4 * H:], views, W3, 0, sc, sh, out=out))
    run(f"node aggregate H={H} E={e}", lambda: ops.node_aggregate(ee, P[:, :H], P[:, H:2 * H], P[:, 2 * H:3 * H], views, h, 0, sc, sh))
    Wc = torch.randn(5 * H, H, device=dev, generator=gen) / H ** 0.5
    bc = torch.randn(5 * H, device=dev, generator=gen)
    run(f"node projection H={H} N={n}", lambda: ops.linear(h, Wc, bc))
    del ee, out, P, views
    torch.cuda.empty_cache()
