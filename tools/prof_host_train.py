import cProfile, pstats, sys, os, torch
sys.path.insert(0, "/root/repo")
import gnnome_amd
from gnnome_amd import ops
from gnnome_amd.loss import bce_loss
from gnnome_amd.synth import make_graph, random_state_dict
dev = torch.device("cuda", 0)
n, e = 2000, 20000   # a METIS-cluster-sized graph: the step is pure host time
g = make_graph(n, e, seed=1)
views = ops.GraphViews(g["src"].to(dev), g["dst"].to(dev), n)
x = ops.degree_features(views); ef = g["e"].to(dev); y = g["y"].to(dev); pw = g["pos_weight"].to(dev)
m = gnnome_amd.SymGatedGCNModel(2, 2, 128, 16, 8, 64, "batch", dropout=0.2); m.load_state_dict(random_state_dict(128, seed=1)); m.to(dev).train()
opt = torch.optim.Adam(m.parameters(), lr=1e-4)
def step():
    logits = m(views, x, ef); loss = bce_loss(logits.squeeze(-1), y, pw); opt.zero_grad(set_to_none=False); loss.backward(); opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
import time; t0 = time.perf_counter()
for _ in range(10): step()
torch.cuda.synchronize(); print("ms/step", (time.perf_counter() - t0) * 100)
pr = cProfile.Profile(); pr.enable()
for _ in range(5): step()
torch.cuda.synchronize(); pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(18)
