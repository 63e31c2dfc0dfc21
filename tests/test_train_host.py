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
