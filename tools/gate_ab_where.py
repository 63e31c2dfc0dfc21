import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnnome_amd import ops
from gnnome_amd.synth import make_graph
dev = torch.device("cuda", 0)
e, H = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000, 128; n = e // 10
g = make_graph(n, e, seed=1)
views = ops.GraphViews(g["src"].to(dev), g["dst"].to(dev), n)
gen = torch.Generator(device=dev).manual_seed(0)
ee = torch.randn(e, H, device=dev, generator=gen)
P = torch.randn(n, 5 * H, device=dev, generator=gen)
W3 = torch.randn(H, H, device=dev, generator=gen) / H ** 0.5
sc, sh = torch.rand(H, device=dev, generator=gen) * 0.1, torch.randn(H, device=dev, generator=gen)
res = {}
for v in (8, 0):
    ops.set_tuning(0, v)
    o = torch.empty_like(ee)
    ops.edge_gate(ee, P[:, 3 * H:4 * H], P[:, 4 * H:], views, W3, 0, sc, sh, out=o)
    res[v] = o
ops.set_tuning(0, 0)
torch.cuda.synchronize()
d = (res[0] - res[8]).abs()
bad = (d > 1e-4)
print("max", d.max().item(), "bad elements", int(bad.sum()), "of", d.numel())
rows = bad.any(1).nonzero().flatten()
print("bad rows", rows.numel(), "first", rows[:20].tolist(), "row%32 hist", torch.bincount(rows % 32, minlength=32).tolist())
cols = bad.any(0).nonzero().flatten()
print("bad cols", cols.numel(), cols[:40].tolist())
if rows.numel():
    r = rows[0].item(); print("row", r, "v0", res[0][r, :8].tolist(), "v7", res[7][r, :8].tolist())
