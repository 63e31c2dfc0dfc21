#!/usr/bin/env python
"""Diagnostic for tests/test_reference_order.py::test_reference_order_is_needed_and_sufficient_at_hidden_256: where does the reference-order
path leave the torch-CPU oracle's bits on THIS host?  (a) torch CPU nn.Linear against the k-ascending fma chain at K = 256 (MKL's order is
host-dependent), (b) layer 0 of the sensitive model piece by piece: projections, gate input, e', h'."""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import gnnome_amd  # noqa: E402
from gnnome_amd import ops  # noqa: E402
from gnnome_amd.synth import make_graph, random_state_dict  # noqa: E402
from oracle.symgated_oracle import degree_features, model_from_state_dict  # noqa: E402


def chain(A, W, b):
    acc = torch.zeros(A.shape[0], W.shape[0], dtype=torch.float64)
    for k in range(A.shape[1]):
        acc = (acc + A[:, k:k + 1].double() * W[:, k].double().unsqueeze(0)).float().double()
    return (acc + b.double()).float()


g = torch.Generator().manual_seed(0)
for M, K, N in ((100, 256, 256), (6000, 256, 1280), (60000, 256, 256), (60000, 128, 128), (60001, 256, 256)):
    A, W, b = 10 * torch.randn(M, K, generator=g), 0.3 * torch.randn(N, K, generator=g), torch.randn(N, generator=g)
    y = F.linear(A, W, b)
    bad = (y != chain(A, W, b))
    print(f"torch CPU linear vs k-ascending chain M={M} K={K} N={N}: mismatch {bad.float().mean().item():.4f}; rows with a mismatch {bad.any(1).float().mean().item():.4f}"
          f" (first bad row {int(bad.any(1).nonzero()[0]) if bad.any() else -1}, threads {torch.get_num_threads()})", flush=True)

hidden, n, e = 256, 6000, 60_000
gr = make_graph(n, e, seed=2, kind="banded")
x = degree_features(gr["src"], gr["dst"], n)
graph = (gr["src"], gr["dst"], n)
sd = random_state_dict(hidden, seed=4)
for k in ("B_1", "B_2", "B_3"):
    sd[f"gnn.convs.0.{k}.weight"] = sd[f"gnn.convs.0.{k}.weight"] * 0.01
sd["gnn.convs.0.B_3.bias"] = 300.0 * (1 + 0.3 * torch.randn(hidden, generator=torch.Generator().manual_seed(1)))
om64 = model_from_state_dict(sd, dtype=torch.float64).eval()
with torch.no_grad():
    h = om64.linear2_node(torch.relu(om64.linear1_node(x.double())))
    ee = om64.linear2_edge(torch.relu(om64.linear1_edge(gr["e"].double())))
    c = om64.gnn.convs[0]
    pre = c.B_1(h)[gr["src"].long()] + c.B_2(h)[gr["dst"].long()] + c.B_3(ee)
sd["gnn.convs.0.bn_e.running_mean"] = pre.mean(0).float()
sd["gnn.convs.0.bn_e.running_var"] = pre.var(0, unbiased=False).float()
sd["gnn.convs.0.bn_e.weight"] = sd["gnn.convs.0.bn_e.weight"].abs() + 0.5
om = model_from_state_dict(sd).eval()
tr = []
with torch.no_grad():
    om(graph, x, gr["e"], trace=tr)
    c = om.gnn.convs[0]
    h0, e0 = tr[0]
    B1h, B2h, B3e = c.B_1(h0), c.B_2(h0), c.B_3(e0)
    pre32 = B1h[gr["src"].long()] + B2h[gr["dst"].long()] + B3e
dev = torch.device("cuda", 0)
m = gnnome_amd.SymGatedGCNModel(2, 2, hidden, 16, 8, 64, "batch").eval()
m.load_state_dict(sd)
m.to(dev)
m.arithmetic = "reference"
from gnnome_amd import engine  # noqa: E402
prep = engine.prepared_for(m, dev, engine.Prepared)
lw = prep.layers[0]
h0d, e0d = h0.to(dev), e0.to(dev)
P = ops.linear_ref(h0d, lw.Wcat, lw.bcat).cpu()
H = hidden
for name, i, want in (("B1h", 3, B1h), ("B2h", 4, B2h)):
    got = P[:, i * H:(i + 1) * H]
    print(f"{name}: mismatch {(got != want).float().mean().item():.4f} max abs diff {(got - want).abs().max().item():.3e}", flush=True)
views = ops.GraphViews(gr["src"].to(dev), gr["dst"].to(dev), n)
es = ops.gather_rows(e0d, views.srt_eid)
Pd = P.to(dev)
out = ops.edge_gate_ref(es.clone(), Pd[:, 3 * H:4 * H], Pd[:, 4 * H:], views, lw.W3, lw.b3, lw.scale_e, lw.shift_e).cpu()
want_e1 = tr[1][1][views.srt_eid.cpu().long()]
bad = out != want_e1
print(f"e' of layer 0 (edge_gate_ref on the oracle's own inputs): mismatch {bad.float().mean().item():.4f} rows {bad.any(1).float().mean().item():.4f} max abs diff {(out - want_e1).abs().max().item():.3e}", flush=True)
# the B_3 product alone, against torch's on this host and against the chain
B3_chain = chain(e0, c.B_3.weight, c.B_3.bias)
print(f"oracle's B_3(e0) vs the chain: mismatch {(B3e != B3_chain).float().mean().item():.4f} rows {(B3e != B3_chain).any(1).float().mean().item():.4f}")
# torch's eval BatchNorm against fma(x, alpha, beta)
with torch.no_grad():
    bn = c.bn_e(pre32)
alpha = (c.bn_e.weight * (1.0 / torch.sqrt(c.bn_e.running_var + c.bn_e.eps)))
beta = (c.bn_e.bias.double() - c.bn_e.running_mean.double() * alpha.double()).float()
fma = (pre32.double() * alpha.double() + beta.double()).float()
print(f"torch eval BatchNorm vs fma(x, alpha, beta): mismatch {(bn != fma).float().mean().item():.4f} max abs diff {(bn - fma).abs().max().item():.3e}")
