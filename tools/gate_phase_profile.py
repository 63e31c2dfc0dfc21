#!/usr/bin/env python
"""Where a compute wave of the edge-tile kernel spends its cycles (gnnome_debug_gate_profile): per-tile averages of
wait-for-slot / prologue / MFMA loop / x write-back, over all workgroups of one launch at a BASELINE size."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnnome_amd import _lib, ops  # noqa: E402
from gnnome_amd.synth import make_graph  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--hidden", type=int, default=128)
ap.add_argument("--edges", type=int, default=1_000_000)
ap.add_argument("--ablation", type=int, default=0)
ap.add_argument("--variant", type=int, default=0, help="gnnome_set_tuning(0, v): 0 = k_edge_gate_bf, 7 = k_edge_gate_pl (slot 2 = waiting for the other compute waves)")
ap.add_argument("--linear", action="store_true", help="profile the node projection [N,128] -> [N,640] on the same kernel (mode 4) instead of the gate; --edges = rows")
a = ap.parse_args()
dev = torch.device("cuda", 0)
n, e, H = a.edges // 10, a.edges, a.hidden
g = make_graph(n, e, seed=1)
views = ops.GraphViews(g["src"].to(dev), g["dst"].to(dev), n)
gen = torch.Generator(device=dev).manual_seed(0)
ee = torch.randn(e, H, device=dev, generator=gen)
out = torch.empty_like(ee)
P = torch.randn(n, 5 * H, device=dev, generator=gen)
W3 = torch.randn(H, H, device=dev, generator=gen) / H ** 0.5
sc, sh = torch.rand(H, device=dev, generator=gen) * 0.1, torch.randn(H, device=dev, generator=gen)
ops.set_tuning(1, a.ablation)
ops.set_tuning(0, a.variant)
if a.linear:
    hrows = torch.randn(e, H, device=dev, generator=gen)
    Wc, bc, Pout = torch.randn(5 * H, H, device=dev, generator=gen), torch.randn(5 * H, device=dev, generator=gen), torch.empty(e, 5 * H, device=dev)
    launch = lambda: ops.linear(hrows, Wc, bc, out=Pout)  # noqa: E731
else:
    launch = lambda: ops.edge_gate(ee, P[:, 3 * H:4 * H], P[:, 4 * H:], views, W3, 0, sc, sh, out=out)  # noqa: E731
for _ in range(3):
    launch()
prof = torch.zeros(2 * 256 * 8, dtype=torch.int64, device=dev)   # [256 compute-wave records | 256 load-wave records (variant 7)]
lib = _lib.load()
lib.gnnome_debug_gate_profile(prof.data_ptr())
s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
launch()
t.record()
torch.cuda.synchronize()
lib.gnnome_debug_gate_profile(None)
ops.set_tuning(1, 0)
ops.set_tuning(0, 0)
p = prof[:256 * 8].view(256, 8).cpu().double()
lw = prof[256 * 8:].view(256, 8).cpu().double()
tiles = p[:, 4].sum().item()
active = int((p[:, 4] > 0).sum().item())
names = ["wait for slot", "prologue (first fragment)", "MFMA loop", "x write-back"]
ms = s.elapsed_time(t)
print(f"H={H} E={e} ablation={a.ablation}: launch {ms:.4f} ms (with counters), {tiles:.0f} tiles, {tiles / max(active, 1):.1f} per workgroup ({active} active)")
total = 0.0
for k, name in enumerate(names):
    per = p[:, k].sum().item() / tiles
    total += per
    print(f"  {name:28s} {per:8.0f} cycles / tile")
span_c, span_r = p[:, 5].mean().item(), p[:, 6].mean().item()
print(f"  loop span per workgroup: {span_c:.0f} shader cycles in {span_r / 100.0:.1f} us (100 MHz counter) -> shader clock {span_c / (span_r / 100.0) / 1e3:.2f} GHz; "
      f"accounted {total * tiles / 256 / span_c:.0%} of the span")
print(f"  {'sum':28s} {total:8.0f} cycles / tile   -> {total * tiles / 256 / (ms * 1e-3) / 1e9:.2f} GHz if the wave were busy for the whole launch")
if H == 256 and lw[:, 4].sum().item() > 0:   # k_edge_gate_pl256: two load groups, the first load wave's view of the tiles IT handles (one in two)
    t = lw[:, 4].sum().item()
    print("  first load wave, cycles per tile it handles (one in two): "
          + ", ".join(f"{nm} {lw[:, k].sum().item() / t:.0f}" for k, nm in [(2, "wait for compute (done)"), (1, "split + publish the tile after next"),
                                                                            (3, "epilogue + stores"), (5, "issue of the next requests")]))
elif a.variant == 7 and lw[:, 4].sum().item() > 0:
    t = lw[:, 4].sum().item()
    print("  first load wave, cycles per tile IT handles (one in four): "
          + ", ".join(f"{nm} {lw[:, k].sum().item() / t:.0f}" for k, nm in enumerate(["fetch arrival", "split + plane stores", "wait for compute", "epilogue + stores"])))
