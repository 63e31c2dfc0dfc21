#!/usr/bin/env python
"""Per-parameter gradient error of one training step vs the oracle's autograd (GPU box; debugging aid)."""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gnnome_amd  # noqa: E402
from gnnome_amd.features import degree_features  # noqa: E402
from gnnome_amd.synth import make_graph, random_state_dict  # noqa: E402
from oracle.symgated_oracle import OracleModel, bce_loss  # noqa: E402

hidden, n, e = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
dev = torch.device("cuda", 0)
gr = make_graph(n, e, seed=9)
x = degree_features(gr["src"], gr["dst"], n)
sd = random_state_dict(hidden, seed=3)
om = OracleModel(2, 2, hidden, 16, 8, 64, "batch", dropout=0.0)
om.load_state_dict(sd)
om.train()
wl = om((gr["src"], gr["dst"], n), x, gr["e"])
bce_loss(wl, gr["y"], gr["pos_weight"]).backward()
m = gnnome_amd.models.SymGatedGCNModel(2, 2, hidden, 16, 8, 64, "batch", dropout=0.0)
m.load_state_dict(sd)
m.to(dev).train()
got = m((gr["src"], gr["dst"], n), x.to(dev), gr["e"].to(dev))
F.binary_cross_entropy_with_logits(got.squeeze(-1), gr["y"].to(dev), pos_weight=gr["pos_weight"].to(dev)).backward()
om64 = OracleModel(2, 2, hidden, 16, 8, 64, "batch", dropout=0.0)
om64.load_state_dict(sd)
om64 = om64.double().train()
bce_loss(om64((gr["src"], gr["dst"], n), x.double(), gr["e"].double()), gr["y"].double(), gr["pos_weight"].double()).backward()
truth = {k: p.grad for k, p in om64.named_parameters()}
print("max dprob", (torch.sigmoid(got.detach().cpu()) - torch.sigmoid(wl.detach())).abs().max().item())
want = {k: p.grad for k, p in om.named_parameters()}
rows = []
for k, p in m.named_parameters():
    w = want[k]
    err = (p.grad.cpu() - w).abs().max().item()
    t = truth[k]
    e_hip = (p.grad.cpu().double() - t).abs().max().item() / (t.abs().max().item() + 1e-30)
    e_orc = (w.double() - t).abs().max().item() / (t.abs().max().item() + 1e-30)
    rows.append((err / (w.abs().max().item() + 1e-12), err, w.abs().max().item(), k + "   vs fp64: hip %.1e  oracle-fp32 %.1e" % (e_hip, e_orc)))
rows.sort(reverse=True)
big = max(r[2] for r in rows)
for r in rows:
    if r[2] > 1e-6 * big and r[0] > 1e-5:
        print("rel %.2e abs %.2e wantmax %.2e %s" % r)
