"""CPU tests of the training step's HOST logic (gnnome_amd/train.py): the kernel sequence, the hand-written backward
and the BatchNorm buffer updates, run on the checker backend (tests/cpu_ops.py = the C-ABI contracts in torch) and
compared with the golden captured from the reference's own classes under autograd (tests/golden/g3_train_h64.pt)."""
import torch
import torch.nn.functional as F

import cpu_ops
import gnnome_amd
from conftest import load_golden
from gnnome_amd.synth import random_state_dict
from gnnome_amd.train import WholeGraph, train_forward_on


def check_grads(got, want, rtol, floor=2e-6):
    assert set(got) == set(want)
    for k, w in want.items():
        err = (got[k].double() - w.double()).abs().max().item()
        assert err <= rtol * w.abs().max().item() + floor, f"{k}: max abs err {err:.2e} vs max |grad| {w.abs().max().item():.2e}"


def test_whole_graph_step_on_checker_backend_matches_reference_golden_g3():
    g = load_golden("g3_train_h64.pt")
    m = gnnome_amd.models.SymGatedGCNModel(2, 2, 64, 16, 8, 64, "batch", dropout=0.0)
    m.load_state_dict(random_state_dict(64, seed=g["seed"]))
    m.train()
    views = cpu_ops.CpuViews(g["src"], g["dst"], g["num_nodes"])
    logits = train_forward_on(m, WholeGraph(views, cpu_ops), g["x"], g["e"])
    loss = F.binary_cross_entropy_with_logits(logits.squeeze(-1), g["y"], pos_weight=g["pos_weight"])
    loss.backward()
    assert (torch.sigmoid(logits.detach()) - torch.sigmoid(g["logits"])).abs().max().item() < 1e-4
    assert abs(loss.item() - g["loss"].item()) < 1e-5
    check_grads({k: p.grad for k, p in m.named_parameters()}, g["grads"], rtol=1e-3)
    bufs = dict(m.named_buffers())
    for k, want in g["buffers_after"].items():
        assert torch.allclose(bufs[k].float(), want.float(), atol=1e-5, rtol=1e-4), k
    assert bufs["gnn.convs.0.bn_e.num_batches_tracked"].item() == 2 and bufs["gnn.convs.0.bn_h.num_batches_tracked"].item() == 1


def test_layernorm_training_step_on_checker_backend_matches_reference_golden_g8():
    """normalization='layer' (gated_gcn_full.py:40-42) in train mode: per-row statistics, hand-written LayerNorm backward."""
    g = load_golden("g8_layernorm_train_h64.pt")
    sd = {k: v for k, v in random_state_dict(64, seed=g["seed"]).items() if "running_" not in k and "num_batches" not in k}
    m = gnnome_amd.models.SymGatedGCNModel(2, 2, 64, 16, 8, 64, "layer", dropout=0.0)
    m.load_state_dict(sd)
    m.train()
    views = cpu_ops.CpuViews(g["src"], g["dst"], g["num_nodes"])
    logits = train_forward_on(m, WholeGraph(views, cpu_ops), g["x"], g["e"])
    loss = F.binary_cross_entropy_with_logits(logits.squeeze(-1), g["y"], pos_weight=g["pos_weight"])
    loss.backward()
    assert (torch.sigmoid(logits.detach()) - torch.sigmoid(g["logits"])).abs().max().item() < 1e-4
    assert abs(loss.item() - g["loss"].item()) < 1e-5
    check_grads({k: p.grad for k, p in m.named_parameters()}, g["grads"], rtol=1e-3)


def test_layernorm_training_step_at_widths_between_the_built_ones_golden_g12():
    """(96, 48) and (160, 128) with normalization='layer' in train mode: the zero-padded twin (train._padded_step) with LayerNorm over the
    model's own channels, and hidden_edge_scores = 128 through the scorer's backward - loss and every gradient, in the model's own shapes,
    against the reference's autograd."""
    g = load_golden("g12_layernorm_widths.pt")
    for case in g["cases"]:
        sd = {k: v for k, v in random_state_dict(case["hidden"], num_layers=case["layers"], hidden_edge_scores=case["hs"], seed=case["seed"]).items()
              if "running_" not in k and "num_batches" not in k}
        m = gnnome_amd.models.SymGatedGCNModel(2, 2, case["hidden"], 16, case["layers"], case["hs"], "layer", dropout=0.0)
        m.load_state_dict(sd)
        m.train()
        views = cpu_ops.CpuViews(g["src"], g["dst"], g["num_nodes"])
        logits = train_forward_on(m, WholeGraph(views, cpu_ops), g["x"], g["e"])
        loss = F.binary_cross_entropy_with_logits(logits.squeeze(-1), g["y"], pos_weight=g["pos_weight"])
        loss.backward()
        assert (torch.sigmoid(logits.detach()) - torch.sigmoid(case["logits"])).abs().max().item() < 1e-4
        assert abs(loss.item() - case["loss"].item()) < 1e-5
        grads = {k: p.grad for k, p in m.named_parameters()}
        assert all(grads[k].shape == w.shape for k, w in case["grads"].items())
        check_grads(grads, case["grads"], rtol=1e-3)


def _fixed_masks(n, H, layers, p, seed):
    g = torch.Generator().manual_seed(seed)
    return [(torch.rand(n, H, generator=g) >= p).float() / (1.0 - p) for _ in range(layers)]


def test_dropout_training_step_matches_oracle_autograd_with_the_same_masks(monkeypatch):
    """dropout = 0.2 is the reference's default training configuration (configs/hyperparameters.py:29).  The masks are
    random, so both sides are given the SAME ones: the oracle through F.dropout, the step through train.dropout_mask."""
    import oracle.symgated_oracle as osg
    from gnnome_amd import train as gtrain
    from gnnome_amd.synth import make_graph
    n, e, H, L, p = 300, 3000, 64, 3, 0.2
    gr = make_graph(n, e, seed=13)
    x = osg.degree_features(gr["src"], gr["dst"], n)
    sd = random_state_dict(H, num_layers=L, seed=7)
    masks = _fixed_masks(n, H, L, p, seed=99)

    it = iter(masks)
    monkeypatch.setattr(osg.F, "dropout", lambda h, pp, training=True: h * next(it) if training and pp > 0 else h)
    om = osg.OracleModel(2, 2, H, 16, L, 64, "batch", dropout=p)
    om.load_state_dict(sd)
    om.train()
    want = om((gr["src"], gr["dst"], n), x, gr["e"])
    want_loss = osg.bce_loss(want, gr["y"], gr["pos_weight"])
    want_loss.backward()
    monkeypatch.undo()

    it2 = iter(masks)
    monkeypatch.setattr(gtrain, "dropout_mask", lambda rows, cols, pp, device: next(it2).to(device))
    m = gnnome_amd.models.SymGatedGCNModel(2, 2, H, 16, L, 64, "batch", dropout=p)
    m.load_state_dict(sd)
    m.train()
    views = cpu_ops.CpuViews(gr["src"], gr["dst"], n)
    got = train_forward_on(m, WholeGraph(views, cpu_ops), x, gr["e"])
    loss = F.binary_cross_entropy_with_logits(got.squeeze(-1), gr["y"], pos_weight=gr["pos_weight"])
    loss.backward()
    assert (torch.sigmoid(got.detach()) - torch.sigmoid(want.detach())).abs().max().item() < 1e-4
    assert abs(loss.item() - want_loss.item()) < 1e-5
    check_grads({k: q.grad for k, q in m.named_parameters()}, {k: q.grad for k, q in om.named_parameters()}, rtol=2e-3)
    # and the default mask generator: right mean, right support
    monkeypatch.undo()
    mk = gtrain.dropout_mask(2000, 64, 0.2, torch.device("cpu"))
    assert set(mk.unique().tolist()) == {0.0, 1.25} and abs(mk.mean().item() - 1.0) < 0.02


def test_bf16_activation_storage_on_checker_backend_stays_close_to_the_reference_golden_g3():
    """model.activation_storage = "bf16" (xe and dxe rounded to bfloat16 between the kernels, arithmetic fp32): the host logic of
    the option, and its distance from the REFERENCE's fp32 step on golden G3 (2000 edges: little averaging) - loss within 1e-4,
    probabilities within 2e-3, the whole gradient within 3 % in L2 (measured 1.2 %; single tensors up to 14 % of their own
    scale; the fp32 step holds 1e-3), BatchNorm buffers within 1e-3.  The option refuses what it is not built for."""
    import pytest
    g = load_golden("g3_train_h64.pt")
    m = gnnome_amd.models.SymGatedGCNModel(2, 2, 64, 16, 8, 64, "batch", dropout=0.0)
    m.load_state_dict(random_state_dict(64, seed=g["seed"]))
    m.train()
    m.activation_storage = "bf16"
    views = cpu_ops.CpuViews(g["src"], g["dst"], g["num_nodes"])
    logits = train_forward_on(m, WholeGraph(views, cpu_ops), g["x"], g["e"])
    loss = F.binary_cross_entropy_with_logits(logits.squeeze(-1), g["y"], pos_weight=g["pos_weight"])
    loss.backward()
    assert (torch.sigmoid(logits.detach()) - torch.sigmoid(g["logits"])).abs().max().item() < 2e-3
    assert abs(loss.item() - g["loss"].item()) < 1e-4 * abs(g["loss"].item())
    got = {k: p.grad for k, p in m.named_parameters()}
    check_grads(got, g["grads"], rtol=0.25, floor=3e-5)
    num = sum(((got[k] - w).double() ** 2).sum().item() for k, w in g["grads"].items()) ** 0.5
    den = sum((w.double() ** 2).sum().item() for w in g["grads"].values()) ** 0.5
    assert num / den < 3e-2, f"relative L2 distance of the gradient from the reference's {num / den:.2e}"
    bufs = dict(m.named_buffers())
    for k, want in g["buffers_after"].items():
        assert torch.allclose(bufs[k].float(), want.float(), atol=1e-3, rtol=1e-3), k
    # LayerNorm has no bf16-storage path; an unknown storage name is rejected
    ln = gnnome_amd.models.SymGatedGCNModel(2, 2, 64, 16, 2, 64, "layer", dropout=0.0).train()
    ln.activation_storage = "bf16"
    with pytest.raises(ValueError):
        train_forward_on(ln, WholeGraph(views, cpu_ops), g["x"], g["e"])
    m.activation_storage = "fp16"
    with pytest.raises(ValueError):
        train_forward_on(m, WholeGraph(views, cpu_ops), g["x"], g["e"])


def _step_grads(golden, norm, recompute, seed_key="seed"):
    g = load_golden(golden)
    sd = random_state_dict(64, seed=g[seed_key])
    if norm == "layer":
        sd = {k: v for k, v in sd.items() if "running_" not in k and "num_batches" not in k}
    m = gnnome_amd.models.SymGatedGCNModel(2, 2, 64, 16, 8, 64, norm, dropout=0.0)
    m.load_state_dict(sd)
    m.train()
    m.recompute_gate = recompute
    views = cpu_ops.CpuViews(g["src"], g["dst"], g["num_nodes"])
    logits = train_forward_on(m, WholeGraph(views, cpu_ops), g["x"], g["e"])
    F.binary_cross_entropy_with_logits(logits.squeeze(-1), g["y"], pos_weight=g["pos_weight"]).backward()
    return logits.detach(), {k: p.grad.clone() for k, p in m.named_parameters()}, {k: b.clone() for k, b in m.named_buffers()}


def test_recompute_gate_gives_the_same_bits_as_the_stored_activations():
    """model.recompute_gate = True: xe is not kept for the backward, the layer's raw gate is launched again (what fits
    BASELINE configs[4]'s 6.25M-edge shard into a rank's HBM with room to spare).  Same logits, gradients and BatchNorm
    buffers, bit for bit, for both normalizations; and it is refused together with bf16 storage."""
    import pytest
    for golden, norm in (("g3_train_h64.pt", "batch"), ("g8_layernorm_train_h64.pt", "layer")):
        l0, g0, b0 = _step_grads(golden, norm, False)
        l1, g1, b1 = _step_grads(golden, norm, True)
        assert torch.equal(l0, l1)
        assert all(torch.equal(g0[k], g1[k]) for k in g0), norm
        assert all(torch.equal(b0[k], b1[k]) for k in b0), norm
    g = load_golden("g3_train_h64.pt")
    m = gnnome_amd.models.SymGatedGCNModel(2, 2, 64, 16, 8, 64, "batch", dropout=0.0).train()
    m.recompute_gate, m.activation_storage = True, "bf16"
    with pytest.raises(ValueError, match="alternatives"):
        train_forward_on(m, WholeGraph(cpu_ops.CpuViews(g["src"], g["dst"], g["num_nodes"]), cpu_ops), g["x"], g["e"])


def test_training_step_at_widths_between_the_built_ones_matches_oracle_autograd():
    """hidden_features / hidden_edge_scores outside {64,128,256} / {32,64,128} in TRAIN mode (the reference takes any width): the step runs on a
    zero-padded twin of the next built widths whose parameters are differentiable functions of the model's own (train._padded_step) - loss,
    every gradient (in the model's own shapes), the BatchNorm buffers and a second step after an optimizer update against the oracle's autograd."""
    from gnnome_amd.synth import make_graph
    from oracle.symgated_oracle import OracleModel, bce_loss, degree_features
    n, ec = 300, 3000
    gr = make_graph(n, ec, seed=7, kind="banded")
    x = degree_features(gr["src"], gr["dst"], n)
    views = cpu_ops.CpuViews(gr["src"], gr["dst"], n)
    for hidden, hs in ((96, 48), (40, 20), (128, 40)):
        sd = random_state_dict(hidden, num_layers=3, hidden_edge_scores=hs, seed=5)
        om = OracleModel(2, 2, hidden, 16, 3, hs, "batch", dropout=0.0)
        om.load_state_dict(sd)
        om.train()
        m = gnnome_amd.models.SymGatedGCNModel(2, 2, hidden, 16, 3, hs, "batch", dropout=0.0)
        m.load_state_dict(sd)
        m.train()
        opt_o, opt_m = torch.optim.SGD(om.parameters(), lr=0.05), torch.optim.SGD(m.parameters(), lr=0.05)
        for step in range(2):
            opt_o.zero_grad()
            opt_m.zero_grad()
            want = om((gr["src"], gr["dst"], n), x, gr["e"])
            loss_o = bce_loss(want, gr["y"], gr["pos_weight"])
            loss_o.backward()
            got = train_forward_on(m, WholeGraph(views, cpu_ops), x, gr["e"])
            loss_m = F.binary_cross_entropy_with_logits(got.squeeze(-1), gr["y"], pos_weight=gr["pos_weight"])
            loss_m.backward()
            assert got.shape == want.shape and abs(loss_m.item() - loss_o.item()) < 2e-5 * (1 + step)
            assert (torch.sigmoid(got.detach()) - torch.sigmoid(want.detach())).abs().max().item() < 1e-4
            check_grads({k: p.grad for k, p in m.named_parameters()}, {k: p.grad for k, p in om.named_parameters()}, rtol=2e-3 * (1 + step))
            ob, mb = dict(om.named_buffers()), dict(m.named_buffers())
            assert all(mb[k].shape == ob[k].shape and torch.allclose(mb[k].float(), ob[k].float(), atol=1e-5, rtol=1e-4) for k in ob)
            opt_o.step()
            opt_m.step()
        assert list(m.state_dict().keys()) == list(sd.keys()) and all(m.state_dict()[k].shape == v.shape for k, v in sd.items())
    # (LayerNorm models at such widths: test_layernorm_training_step_at_widths_between_the_built_ones_golden_g12)
