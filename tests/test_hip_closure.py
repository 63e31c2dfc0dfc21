"""GPU tests of the callers' closure: feature preparation, losses and confusion counts on
the device, against the golden taken from the reference's own functions (g7_closure.pt) and against the oracle at
larger sizes."""
import pytest
import torch

import gnnome_amd
from conftest import load_golden
from gnnome_amd import features, loss, metrics, ops
from gnnome_amd.synth import make_graph
from oracle import symgated_oracle as oracle

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda", 0)


def test_losses_and_gradients_match_reference_golden_g7():
    g = load_golden("g7_closure.pt")
    labels, pw = g["labels"].to(dev()), g["pos_weight"].to(dev())
    org, rev = g["org"].to(dev()).requires_grad_(), g["rev"].to(dev()).requires_grad_()
    sym = loss.symmetry_loss(org, rev, labels, pw, alpha=g["alpha"])
    sym.backward()
    assert abs(sym.item() - g["symmetry_loss"].item()) <= 2e-6 * abs(g["symmetry_loss"].item())
    # per-edge gradients are O(1/E); compare at their own scale (device expf / CPU expf differ in the last ulp)
    scale = g["symmetry_grad_org"].abs().max().item()
    assert (org.grad.cpu() - g["symmetry_grad_org"]).abs().max().item() <= 2e-6 * scale
    assert (rev.grad.cpu() - g["symmetry_grad_rev"]).abs().max().item() <= 2e-6 * scale
    # ties |a-b| = 0 take the zero subgradient, exactly as torch.abs
    assert torch.equal(org.grad[:100].cpu() - g["symmetry_grad_org"][:100], rev.grad[:100].cpu() - g["symmetry_grad_rev"][:100])
    org.grad = None
    bce = loss.bce_loss(org, labels, pw)
    (2.0 * bce).backward()   # an upstream factor must reach the logits
    assert abs(bce.item() - g["bce_loss"].item()) <= 2e-6 * abs(g["bce_loss"].item())
    assert (org.grad.cpu() - 2.0 * g["bce_grad"]).abs().max().item() <= 4e-6 * g["bce_grad"].abs().max().item()
    # python-float pos_weight and the defaults of train.py:103
    got = loss.symmetry_loss(org.detach(), rev.detach(), labels)
    want = oracle.symmetry_loss(g["org"], g["rev"], g["labels"], torch.tensor(1.0), 1.0)
    assert abs(got.item() - want.item()) <= 2e-6 * want.item()


def test_confusion_counts_and_scores_match_reference_golden_g7():
    g = load_golden("g7_closure.pt")
    counts = metrics.calculate_tfpn(g["org"].to(dev()), g["labels"].to(dev()))
    assert counts == tuple(g["tfpn"]) and all(isinstance(c, int) for c in counts)
    assert metrics.calculate_tfpn(g["rev"].to(dev()).unsqueeze(1), g["labels"].to(dev())) == tuple(g["tfpn_rev"])
    assert metrics.calculate_metrics(*counts) == pytest.approx(g["metrics"], rel=1e-12)
    assert metrics.calculate_metrics_inverse(*counts) == pytest.approx(g["metrics_inverse"], rel=1e-12)
    assert metrics.calculate_metrics(0, 5, 0, 0) == (1.0, 0, 0, 0)   # the reference's ZeroDivisionError branches


def test_feature_preparation_matches_reference_golden_g7():
    g = load_golden("g7_closure.pt")
    views = gnnome_amd.graph.views_for((g["src"], g["dst"], g["num_nodes"]), dev())
    x = features.degree_features_hip(views)
    assert (x.cpu() - g["x"]).abs().max().item() <= 2e-6
    assert (features.degree_features_hip(views, reverse=True).cpu() - g["x_reversed"]).abs().max().item() <= 2e-6
    # train.py:165-166, literally: g = dgl.reverse(g, True, True); get_full_ne_features(g, reverse=True).  The degrees are
    # stored features that dgl.reverse copies, so only the flag swaps the columns
    assert torch.equal(features.degree_features_hip(views.reversed(), reverse=True), features.degree_features_hip(views, reverse=True))
    assert torch.equal(features.degree_features_hip(views.reversed()), x)

    class _Stored:   # a GNNome DGLGraph carries the degrees as ndata (graph_parser.py); reversed graphs keep the copies
        ndata = {"in_deg": torch.bincount(g["dst"].long(), minlength=g["num_nodes"]), "out_deg": torch.bincount(g["src"].long(), minlength=g["num_nodes"])}
    assert (features.degree_features(_Stored(), device=dev()).cpu() - g["x"]).abs().max().item() <= 2e-6
    assert (features.degree_features(_Stored(), reverse=True, device=dev()).cpu() - g["x_reversed"]).abs().max().item() <= 2e-6
    e = features.edge_features_hip(g["overlap_length"].to(dev()), g["overlap_similarity"].to(dev()))
    assert (e.cpu() - g["e"]).abs().max().item() <= 2e-6 and torch.equal(e[:, 1].cpu(), g["overlap_similarity"])


def test_full_size_against_the_oracle_and_determinism():
    n, e = 100_000, 1_000_000
    gr = make_graph(n, e, seed=1)
    views = gnnome_amd.graph.views_for((gr["src"], gr["dst"], n), dev())
    x = ops.degree_features(views)
    assert (x.cpu() - oracle.degree_features(gr["src"], gr["dst"], n)).abs().max().item() <= 5e-6
    gen = torch.Generator().manual_seed(5)
    a, b = 4.0 * torch.randn(e, generator=gen), 4.0 * torch.randn(e, generator=gen)
    want = oracle.symmetry_loss(a.double(), b.double(), gr["y"].double(), gr["pos_weight"].double(), 0.3)
    got = [ops.edge_loss(a.to(dev()), b.to(dev()), gr["y"].to(dev()), gr["pos_weight"].to(dev()), 0.3, need_counts=True) for _ in range(2)]
    assert abs(got[0][0].item() - want.item()) <= 1e-6 * want.item()
    assert all(torch.equal(p, q) for p, q in zip(got[0], got[1]))          # same bits on every launch
    assert tuple(got[0][3].tolist()) == oracle.calculate_tfpn(a, gr["y"])
    lens = torch.randint(500, 40_000, (e,), generator=gen)
    sims = torch.rand(e, generator=gen)
    feat = ops.edge_features(lens.to(dev()), sims.to(dev()))
    assert (feat.cpu() - oracle.edge_features(lens, sims)).abs().max().item() <= 5e-6
