"""The N>1 path on CPU: destination-range partition, halo plan and per-layer exchange over gloo, with the
checker backend as compute, must reproduce the single-process oracle on every rank."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

from conftest import load_golden
from dist_worker import worker
from gnnome_amd.dist import split_by_incident_edges
from oracle.symgated_oracle import degree_features
from gnnome_amd.synth import make_graph, random_state_dict
from oracle.symgated_oracle import model_from_state_dict


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(world, case, tmp_path):
    path = os.path.join(tmp_path, "case.pt")
    torch.save(case, path)
    mp.spawn(worker, args=(world, _free_port(), path, str(tmp_path)), nprocs=world, join=True)
    return [torch.load(os.path.join(tmp_path, f"rank{r}.pt"), weights_only=False) for r in range(world)]


@pytest.mark.parametrize("world", [2, 3])
def test_partitioned_matches_oracle_on_golden_g2(world, tmp_path, shipped_weights):
    g = load_golden("g2_uniform_1k.pt")  # uniform graph: ~ (world-1)/world of the edges are cut
    case = dict(src=g["src"], dst=g["dst"], num_nodes=g["num_nodes"], x=g["x"], e=g["e"], hidden=64, layers=8,
                state_dict=shipped_weights)
    outs = _run(world, case, tmp_path)
    want = g["logits"].squeeze(1)
    for o in outs:
        assert o["logits"].shape == want.shape
        assert (torch.sigmoid(o["logits"]) - torch.sigmoid(want)).abs().max().item() < 1e-4
    assert torch.equal(outs[0]["logits"], outs[-1]["logits"])              # every rank holds the full result
    assert sum(o["n_own"] for o in outs) == g["num_nodes"]
    assert sum(o["n_score"] for o in outs) == g["src"].numel()             # each edge scored exactly once
    assert all(sum(o["send"]) > 0 and sum(o["recv"]) == o["n_local"] - o["n_own"] for o in outs)


def test_partitioned_banded_graph_has_small_halo(tmp_path):
    n, e = 4000, 40000
    gr = make_graph(n, e, seed=2, kind="banded")
    sd = random_state_dict(64, num_layers=3, seed=1)
    x = degree_features(gr["src"], gr["dst"], n)
    case = dict(src=gr["src"], dst=gr["dst"], num_nodes=n, x=x, e=gr["e"], hidden=64, layers=3, state_dict=sd)
    outs = _run(2, case, tmp_path)
    with torch.no_grad():
        want = model_from_state_dict(sd).eval()((gr["src"], gr["dst"], n), x, gr["e"]).squeeze(1)
    for o in outs:
        assert (torch.sigmoid(o["logits"]) - torch.sigmoid(want)).abs().max().item() < 1e-4
    # layout-ordered graph: only the range boundary and the 1 % long-range edges are cut
    assert sum(o["e_local"] for o in outs) < 1.1 * e
    assert all(o["n_local"] - o["n_own"] < 0.1 * n for o in outs)


def test_split_balances_incident_edges():
    gr = make_graph(10000, 100000, seed=4, kind="banded")
    b = split_by_incident_edges(gr["src"].long(), gr["dst"].long(), 10000, 8)
    assert b[0] == 0 and b[-1] == 10000 and all(b[i] <= b[i + 1] for i in range(8))
    deg = torch.bincount(gr["src"].long(), minlength=10000) + torch.bincount(gr["dst"].long(), minlength=10000)
    loads = [int(deg[b[i]:b[i + 1]].sum()) for i in range(8)]
    assert max(loads) < 1.05 * (sum(loads) / 8)
    assert split_by_incident_edges(torch.zeros(0, dtype=torch.long), torch.zeros(0, dtype=torch.long), 5, 2)[-1] == 5


@pytest.mark.parametrize("world", [2, 3])
def test_partitioned_training_step_matches_reference_golden_g3(world, tmp_path):
    """SURVEY.md 8e "Training additions": BatchNorm statistics over the whole graph with every edge counted once,
    halo gradients returned to their owners, parameter gradients summed over ranks - against the golden taken from
    the reference's classes under autograd on the UNPARTITIONED graph."""
    from test_train_host import check_grads
    g = load_golden("g3_train_h64.pt")
    case = dict(src=g["src"], dst=g["dst"], num_nodes=g["num_nodes"], x=g["x"], e=g["e"], y=g["y"], pos_weight=g["pos_weight"],
                hidden=64, layers=8, state_dict=random_state_dict(64, seed=g["seed"]), train=True)
    outs = _run(world, case, tmp_path)
    assert sum(o["n_score"] for o in outs) == g["src"].numel() and sum(o["e_local"] for o in outs) > 1.05 * g["src"].numel()
    for o in outs:
        assert (torch.sigmoid(o["logits"]) - torch.sigmoid(g["logits"].squeeze(-1))).abs().max().item() < 1e-4
        assert abs(o["loss"].item() - g["loss"].item()) < 1e-5
        check_grads(o["grads"], g["grads"], rtol=1e-3)
        for k, want in g["buffers_after"].items():
            assert torch.allclose(o["buffers"][k].float(), want.float(), atol=1e-5, rtol=1e-4), k
    for k in outs[0]["grads"]:   # identical on every rank: optimizers stay in step without a broadcast
        assert torch.equal(outs[0]["grads"][k], outs[-1]["grads"][k]), k


def test_partitioned_training_on_a_mostly_cut_graph_matches_oracle_autograd(tmp_path):
    """uniform graph at world 3: two thirds of the edges are replicated on two ranks, so nearly every gradient row is
    assembled from partial sums on different ranks."""
    from oracle.symgated_oracle import OracleModel, bce_loss
    from test_train_host import check_grads
    n, e = 400, 4000
    gr = make_graph(n, e, seed=4, kind="uniform")
    x = degree_features(gr["src"], gr["dst"], n)
    sd = random_state_dict(64, num_layers=3, seed=2)
    om = OracleModel(2, 2, 64, 16, 3, 64, "batch", dropout=0.0)
    om.load_state_dict(sd)
    om.train()
    want = om((gr["src"], gr["dst"], n), x, gr["e"])
    want_loss = bce_loss(want, gr["y"], gr["pos_weight"])
    want_loss.backward()
    case = dict(src=gr["src"], dst=gr["dst"], num_nodes=n, x=x, e=gr["e"], y=gr["y"], pos_weight=gr["pos_weight"], hidden=64, layers=3,
                state_dict=sd, train=True)
    outs = _run(3, case, tmp_path)
    assert sum(o["e_local"] for o in outs) > 1.5 * e
    for o in outs:
        assert (torch.sigmoid(o["logits"]) - torch.sigmoid(want.detach().squeeze(-1))).abs().max().item() < 1e-4
        assert abs(o["loss"].item() - want_loss.item()) < 1e-5
        check_grads(o["grads"], {k: p.grad for k, p in om.named_parameters()}, rtol=2e-3)
        for k, b in om.named_buffers():
            assert torch.allclose(o["buffers"][k].float(), b.float(), atol=1e-5, rtol=1e-4), k


def test_partitioned_training_step_at_h256_matches_oracle_autograd(tmp_path):
    """The width of BASELINE configs[3] / [4]: at H = 256 the host takes the UNFUSED backward sequence (bn_bwd_apply with the
    mean terms on the owned rows only + a separate data-gradient GEMM) - here on the checker backend over gloo, the GPU twin
    is tests/test_hip_partition.py::test_two_ranks_training_step_h256_matches_oracle_autograd."""
    from oracle.symgated_oracle import OracleModel, bce_loss
    from test_train_host import check_grads
    n, e, hidden, layers = 300, 3000, 256, 2
    gr = make_graph(n, e, seed=6, kind="banded")
    x = degree_features(gr["src"], gr["dst"], n)
    sd = random_state_dict(hidden, num_layers=layers, seed=7)
    om = OracleModel(2, 2, hidden, 16, layers, 64, "batch", dropout=0.0)
    om.load_state_dict(sd)
    om.train()
    want = om((gr["src"], gr["dst"], n), x, gr["e"])
    want_loss = bce_loss(want, gr["y"], gr["pos_weight"])
    want_loss.backward()
    case = dict(src=gr["src"], dst=gr["dst"], num_nodes=n, x=x, e=gr["e"], y=gr["y"], pos_weight=gr["pos_weight"], hidden=hidden,
                layers=layers, state_dict=sd, train=True)
    outs = _run(2, case, tmp_path)
    assert sum(o["n_score"] for o in outs) == e and sum(o["e_local"] for o in outs) > e
    for o in outs:
        assert (torch.sigmoid(o["logits"]) - torch.sigmoid(want.detach().squeeze(-1))).abs().max().item() < 1e-4
        assert abs(o["loss"].item() - want_loss.item()) < 1e-5
        check_grads(o["grads"], {k: p.grad for k, p in om.named_parameters()}, rtol=2e-3)
    for k in outs[0]["grads"]:
        assert torch.equal(outs[0]["grads"][k], outs[1]["grads"][k]), k


def test_partitioned_layernorm_training_step_matches_reference_golden_g8(tmp_path):
    """normalization='layer': per-row statistics need no cross-rank reduction; the partial gradients of replicated
    edges still add up because the LayerNorm backward is linear in the incoming gradient."""
    from test_train_host import check_grads
    g = load_golden("g8_layernorm_train_h64.pt")
    sd = {k: v for k, v in random_state_dict(64, seed=g["seed"]).items() if "running_" not in k and "num_batches" not in k}
    case = dict(src=g["src"], dst=g["dst"], num_nodes=g["num_nodes"], x=g["x"], e=g["e"], y=g["y"], pos_weight=g["pos_weight"],
                hidden=64, layers=8, state_dict=sd, train=True, normalization="layer")
    outs = _run(2, case, tmp_path)
    for o in outs:
        assert (torch.sigmoid(o["logits"]) - torch.sigmoid(g["logits"].squeeze(-1))).abs().max().item() < 1e-4
        assert abs(o["loss"].item() - g["loss"].item()) < 1e-5
        check_grads(o["grads"], g["grads"], rtol=1e-3)
