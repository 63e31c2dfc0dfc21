import os, sys, torch
sys.path.insert(0, "/root/repo")
from gnnome_amd import ops
from gnnome_amd.synth import make_graph
dev = torch.device("cuda", 0)
e, H = 1_000_000, 128; n = e // 10
g = make_graph(n, e, seed=1)
views = ops.GraphViews(g["src"].to(dev), g["dst"].to(dev), n)
gen = torch.Generator(device=dev).manual_seed(0)
ee = torch.randn(e, H, device=dev, generator=gen); out = torch.empty_like(ee)
P = torch.randn(n, 5 * H, device=dev, generator=gen)
W3 = torch.randn(H, H, device=dev, generator=gen) / H ** 0.5
sc, sh = torch.rand(H, device=dev, generator=gen) * 0.1, torch.randn(H, device=dev, generator=gen)
res = {}
for v in (8, 0):
    ops.set_tuning(0, v)
    o = torch.empty_like(ee)
    ops.edge_gate(ee, P[:, 3 * H:4 * H], P[:, 4 * H:], views, W3, 0, sc, sh, out=o)
    res[v] = o
torch.cuda.synchronize()
print("max|plane form - second generation|", (res[0] - res[8]).abs().max().item(), "scale", res[0].abs().max().item())
for rnd in range(3):
    for v in (8, 0):
        ops.set_tuning(0, v)
        for _ in range(5):
            ops.edge_gate(ee, P[:, 3 * H:4 * H], P[:, 4 * H:], views, W3, 0, sc, sh, out=out)
        s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(100):
            ops.edge_gate(ee, P[:, 3 * H:4 * H], P[:, 4 * H:], views, W3, 0, sc, sh, out=out)
        t.record(); torch.cuda.synchronize()
        print(f"round {rnd} variant {v}: {s.elapsed_time(t) / 100:.4f} ms")
ops.set_tuning(0, 0)
