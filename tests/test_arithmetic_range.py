"""fp16x3 has fp16's operand range; the reference (gated_gcn_full.py:97, a plain fp32 nn.Linear) has fp32's.  The model gives the
caller the reference's domain: a forward whose activations or weights leave fp16's range comes back finite and within the 1e-4 bar of
the oracle, without the caller touching gnnome_set_tuning (ADVICE r4 on the scorer)."""
import pytest
import torch

import gnnome_amd
from gnnome_amd import engine, ops
from gnnome_amd.synth import make_graph, random_state_dict
from oracle.symgated_oracle import degree_features, model_from_state_dict

pytestmark = pytest.mark.gpu

DEV = torch.device("cuda", 0)


def _case(hidden, layers, seed, n=3000, e=30000):
    g = make_graph(n, e, seed, "banded")
    x = degree_features(g["src"], g["dst"], n)
    return g, x, random_state_dict(hidden, num_layers=layers, seed=seed)


def _model(sd, hidden, layers):
    m = gnnome_amd.SymGatedGCNModel(2, 2, hidden, 16, layers, 64, "batch").eval()
    m.load_state_dict(sd)
    return m.to(DEV)


def _oracle_prob(sd, g, x):
    with torch.no_grad():
        return torch.sigmoid(model_from_state_dict(sd).eval()((g["src"], g["dst"], g["num_nodes"]), x, g["e"]))


@pytest.mark.parametrize("hidden", [128, 256])
def test_activations_that_cross_fp16s_range_mid_stack(hidden):
    """Layer 1's bn_e gain is raised until |e| passes 65504 from that layer on (checked on the oracle).  The fp16x3 kernels answer with
    NaN rows (range_check off shows them); the default call returns what the fp32 oracle returns, and set_tuning's state is untouched."""
    layers = 4
    g, x, sd = _case(hidden, layers, seed=3)
    sd["gnn.convs.1.bn_e.weight"] = sd["gnn.convs.1.bn_e.weight"] * 4.0e4
    sd["predictor.W1.weight"] = sd["predictor.W1.weight"].clone()
    sd["predictor.W1.weight"][:, 2 * hidden:] *= 2.0e-6      # (the scorer sees e through small weights: the logits stay of order one and the 1e-4 bar on probabilities means something)
    trace = []
    with torch.no_grad():
        model_from_state_dict(sd).eval()((g["src"], g["dst"], g["num_nodes"]), x, g["e"], trace=trace)
    tops = [float(ee.abs().max()) for _, ee in trace[1:]]     # |e| after every layer
    assert tops[0] < ops.FP16_MAX < tops[1], tops     # in range after layer 0, out of it from layer 1 on
    want = _oracle_prob(sd, g, x)
    assert 0.005 < float(want.std())                   # not saturated
    m = _model(sd, hidden, layers)
    xd, ed = x.to(DEV), g["e"].to(DEV)
    graph = (g["src"], g["dst"], g["num_nodes"])
    assert ops._TUNING.get(10, 0) == 0
    m.range_check = False
    raw = m(graph, xd, ed)
    assert not torch.isfinite(raw).all()              # loud: NaN rows, never wrong finite values
    m.range_check = True
    got = m(graph, xd, ed)
    assert torch.isfinite(got).all()
    assert (torch.sigmoid(got).cpu() - want).abs().max().item() < 1e-4
    assert ops._TUNING.get(10, 0) == 0                # the caller's arithmetic selection is as it was
    prep = m.__dict__["_gnnome_prepared"][1]
    assert not prep.force_bf16x6
    # the same tensors on cached views again: remembered, straight to bf16x6 (and the same bits)
    views = gnnome_amd.graph.views_for(type("G", (), {"edges": lambda s: (g["src"], g["dst"]), "num_nodes": lambda s: g["num_nodes"]})(), DEV)
    a, b = m(views, xd, ed), m(views, xd, ed)
    assert torch.equal(a, b) and torch.isfinite(a).all() and engine._same_inputs(prep.range_failed, views, xd, ed)
    # inputs in range: verified once, fp16x3 kept
    sd_ok = random_state_dict(hidden, num_layers=layers, seed=3)
    m2 = _model(sd_ok, hidden, layers)
    c = m2(views, xd, ed)
    prep2 = m2.__dict__["_gnnome_prepared"][1]
    assert engine._same_inputs(prep2.range_verified, views, xd, ed) and prep2.range_failed is None
    assert (torch.sigmoid(c).cpu() - _oracle_prob(sd_ok, g, x)).abs().max().item() < 1e-4
    xd2 = xd.clone()
    xd2 += 0.0                                           # (an in-place change: another version, checked again)
    assert not engine._same_inputs(prep2.range_verified, views, xd2, ed)


def test_weights_outside_fp16s_range_run_as_bf16x6_from_the_start():
    """A weight beyond 65504 (B_3 scaled up, bn_e's gain scaled down by the same factor so that the activations stay ordinary)."""
    hidden, layers = 128, 3
    g, x, sd = _case(hidden, layers, seed=5)
    sd["gnn.convs.1.B_3.weight"] = sd["gnn.convs.1.B_3.weight"] * 2.0e6
    sd["gnn.convs.1.B_3.bias"] = sd["gnn.convs.1.B_3.bias"] * 2.0e6
    sd["gnn.convs.1.B_1.weight"], sd["gnn.convs.1.B_1.bias"] = sd["gnn.convs.1.B_1.weight"] * 2.0e6, sd["gnn.convs.1.B_1.bias"] * 2.0e6
    sd["gnn.convs.1.B_2.weight"], sd["gnn.convs.1.B_2.bias"] = sd["gnn.convs.1.B_2.weight"] * 2.0e6, sd["gnn.convs.1.B_2.bias"] * 2.0e6
    sd["gnn.convs.1.bn_e.running_mean"] = sd["gnn.convs.1.bn_e.running_mean"] * 2.0e6
    sd["gnn.convs.1.bn_e.running_var"] = sd["gnn.convs.1.bn_e.running_var"] * 4.0e12
    want = _oracle_prob(sd, g, x)
    m = _model(sd, hidden, layers)
    m.arithmetic = "fast"      # (the layer's gain is ordinary; keep the reference-order kernels out of this test)
    got = m((g["src"], g["dst"], g["num_nodes"]), x.to(DEV), g["e"].to(DEV))
    assert m.__dict__["_gnnome_prepared"][1].force_bf16x6
    assert torch.isfinite(got).all() and (torch.sigmoid(got).cpu() - want).abs().max().item() < 1e-4
    assert ops._TUNING.get(10, 0) == 0


@pytest.mark.parametrize("hidden", [64, 128])
def test_scorer_keeps_an_out_of_range_operand_loud(hidden):
    """the fp16x3 scorer's two relus turned NaN / -inf into 0 and returned a finite, wrong logit."""
    n, e, hs = 300, 2100, 64
    gen = torch.Generator().manual_seed(1)
    src, dst = torch.randint(0, n, (e,), generator=gen).int(), torch.randint(0, n, (e,), generator=gen).int()
    views = ops.GraphViews(src.to(DEV), dst.to(DEV), n)
    ee = torch.randn(e, hidden, generator=gen).to(DEV)
    PQ = torch.randn(n, 2 * hs, generator=gen).to(DEV)
    W1e = (torch.randn(hs, hidden, generator=gen) / hidden ** 0.5).to(DEV)
    W2, b2 = (torch.randn(32, hs, generator=gen) / 8).to(DEV), torch.randn(32, generator=gen).to(DEV)
    W3, b3 = torch.randn(32, generator=gen).to(DEV), torch.randn(1, generator=gen).to(DEV)
    logits = torch.empty(e, device=DEV)
    ops.edge_score(ee, PQ[:, :hs], PQ[:, hs:], views, W1e, W2, b2, W3, b3, logits, scatter_to_edge_id=False)
    clean = logits.clone()
    ee[11, 5] = 1.0e5        # beyond fp16's range
    ee[40, 7] = -3.0e5
    ops.edge_score(ee, PQ[:, :hs], PQ[:, hs:], views, W1e, W2, b2, W3, b3, logits, scatter_to_edge_id=False)
    assert not torch.isfinite(logits[11]) and not torch.isfinite(logits[40])
    keep = torch.ones(e, dtype=torch.bool, device=DEV)
    keep[[11, 40]] = False
    assert torch.equal(logits[keep], clean[keep])
    with ops.bf16x6_arithmetic():   # fp32's range
        ops.edge_score(ee, PQ[:, :hs], PQ[:, hs:], views, W1e, W2, b2, W3, b3, logits, scatter_to_edge_id=False)
    assert torch.isfinite(logits).all()
