"""GPU tests of the partitioned (one process per GPU) path with the HIP kernels as compute.  The test box has ONE
MI355X, so the ranks are two processes sharing it, and the collectives go over gloo with host staging
(gnnome_amd.dist._staged) instead of RCCL over xGMI: everything except the transport is the multi-GPU product path -
partition, halo plan, per-layer exchange, owned-prefix scoring, and for training the cross-rank BatchNorm
statistics, halo-gradient scatter-add and gradient sum."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

from conftest import load_golden
from dist_worker import worker
from gnnome_amd import ops
from gnnome_amd.synth import random_state_dict
from test_train_host import check_grads

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(world, case, tmp_path):
    path = os.path.join(tmp_path, "case.pt")
    torch.save(case, path)
    mp.spawn(worker, args=(world, _free_port(), path, str(tmp_path)), nprocs=world, join=True)
    return [torch.load(os.path.join(tmp_path, f"rank{r}.pt"), weights_only=False) for r in range(world)]


def test_scatter_add_rows_is_the_transpose_of_gather_rows():
    dev = torch.device("cuda", 0)
    gen = torch.Generator().manual_seed(0)
    out = torch.randn(5000, 128, generator=gen)
    idx = torch.randperm(5000, generator=gen)[:1234].int()
    src = torch.randn(1234, 128, generator=gen)
    want = out.clone()
    want[idx.long()] += src
    got = out.to(dev)
    ops.scatter_add_rows(src.to(dev), idx.to(dev), got)
    assert torch.equal(got.cpu(), want)
    # strided destination (a column block of a wider table)
    wide = torch.zeros(5000, 256, device=dev)
    ops.scatter_add_rows(src.to(dev), idx.to(dev), wide[:, 128:])
    assert torch.equal(wide[:, 128:].cpu()[idx.long()], src) and wide[:, :128].abs().sum().item() == 0


def test_two_ranks_inference_matches_reference_golden_g2(tmp_path, shipped_weights):
    g = load_golden("g2_uniform_1k.pt")
    case = dict(src=g["src"], dst=g["dst"], num_nodes=g["num_nodes"], x=g["x"], e=g["e"], hidden=64, layers=8,
                state_dict=shipped_weights, device="cuda")
    outs = _run(2, case, tmp_path)
    for o in outs:
        assert (torch.sigmoid(o["logits"]) - torch.sigmoid(g["logits"].squeeze(1))).abs().max().item() < 1e-4
    assert torch.equal(outs[0]["logits"], outs[1]["logits"])
    assert sum(o["n_score"] for o in outs) == g["src"].numel() and all(sum(o["send"]) > 0 for o in outs)


def test_two_ranks_training_step_matches_reference_golden_g3(tmp_path):
    g = load_golden("g3_train_h64.pt")
    case = dict(src=g["src"], dst=g["dst"], num_nodes=g["num_nodes"], x=g["x"], e=g["e"], y=g["y"], pos_weight=g["pos_weight"],
                hidden=64, layers=8, state_dict=random_state_dict(64, seed=g["seed"]), train=True, device="cuda")
    outs = _run(2, case, tmp_path)
    for o in outs:
        assert (torch.sigmoid(o["logits"]) - torch.sigmoid(g["logits"].squeeze(-1))).abs().max().item() < 1e-4
        assert abs(o["loss"].item() - g["loss"].item()) < 1e-5
        check_grads(o["grads"], g["grads"], rtol=1e-3)
        for k, want in g["buffers_after"].items():
            assert torch.allclose(o["buffers"][k].float(), want.float(), atol=1e-5, rtol=1e-4), k
    for k in outs[0]["grads"]:
        assert torch.equal(outs[0]["grads"][k], outs[1]["grads"][k]), k
