import sys, time, torch
sys.path.insert(0, "/root/repo")
from gnnome_amd import ops
from gnnome_amd.synth import make_graph
dev = torch.device("cuda", 0)
for i in range(6):
    g = make_graph(100_000, 1_000_000, seed=10 + i)
    src, dst = g["src"].to(dev), g["dst"].to(dev)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    v = ops.GraphViews(src, dst, 100_000, validate="lazy")
    x = ops.degree_features(v)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print(f"graph {i}: views + degree features {1e3 * (t1 - t0):.2f} ms")
