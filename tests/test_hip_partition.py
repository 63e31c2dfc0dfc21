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


@pytest.mark.parametrize("world", [1, 2])
def test_partitioned_forward_replayed_from_hipgraph_segments(tmp_path, world):
    """dist.CapturedPartitionedForward: the kernels between two collectives recorded as one hipGraph each, the halo
    exchanges and the logits all-gather issued between the replays - bit-identical to the eager partitioned forward, on one
    rank and on two ranks sharing the GPU (banded 40k-edge graph, H = 128: cut edges, halo rows both ways)."""
    from gnnome_amd.synth import make_graph
    from oracle.symgated_oracle import degree_features
    n, e, hidden = 4000, 40_000, 128
    gr = make_graph(n, e, seed=4, kind="banded")
    case = dict(src=gr["src"], dst=gr["dst"], num_nodes=n, x=degree_features(gr["src"], gr["dst"], n), e=gr["e"], hidden=hidden, layers=8,
                state_dict=random_state_dict(hidden, seed=4), device="cuda", captured=True)
    outs = _run(world, case, tmp_path)
    for o in outs:
        assert o["replayed"] is not None and torch.equal(o["replayed"], o["logits"])
    assert all(torch.equal(o["logits"], outs[0]["logits"]) for o in outs)
    if world > 1:
        assert all(sum(o["send"]) > 0 for o in outs)


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


@pytest.mark.parametrize("hidden", [128, 256])
def test_two_ranks_at_the_widths_of_configs_3_and_4(tmp_path, hidden):
    """BASELINE configs[3] / [4] run at H = 256 (and the 10M-edge target at H = 128): the partitioned forward against the
    reference goldens G5 at those widths, both ranks bit-identical."""
    g = load_golden(f"g5_eval_h{hidden}.pt")
    case = dict(src=g["src"], dst=g["dst"], num_nodes=g["num_nodes"], x=g["x"], e=g["e"], hidden=hidden, layers=8,
                state_dict=random_state_dict(hidden, seed=g["seed"]), device="cuda")
    outs = _run(2, case, tmp_path)
    for o in outs:
        assert (torch.sigmoid(o["logits"]) - torch.sigmoid(g["logits"].squeeze(1))).abs().max().item() < 1e-4
    assert torch.equal(outs[0]["logits"], outs[1]["logits"])
    assert sum(o["n_score"] for o in outs) == g["src"].numel()


def test_two_ranks_h256_banded_200k_edges_equals_one_rank(tmp_path):
    """A layout-ordered graph at H = 256 large enough that every kernel runs many tiles per rank: two ranks against the
    single-process forward (same kernels, different row sets and halo traffic)."""
    import gnnome_amd
    from gnnome_amd.synth import make_graph
    from oracle.symgated_oracle import degree_features
    n, e, hidden = 20_000, 200_000, 256
    gr = make_graph(n, e, seed=3, kind="banded")
    x = degree_features(gr["src"], gr["dst"], n)
    sd = random_state_dict(hidden, seed=2)
    case = dict(src=gr["src"], dst=gr["dst"], num_nodes=n, x=x, e=gr["e"], hidden=hidden, layers=8, state_dict=sd, device="cuda")
    outs = _run(2, case, tmp_path)
    m = gnnome_amd.SymGatedGCNModel(2, 2, hidden, 16, 8, 64, "batch").eval()
    m.load_state_dict(sd)
    dev = torch.device("cuda", 0)
    want = m.to(dev)((gr["src"], gr["dst"], n), x.to(dev), gr["e"].to(dev)).squeeze(1).cpu()
    for o in outs:
        assert (torch.sigmoid(o["logits"]) - torch.sigmoid(want)).abs().max().item() < 1e-5
    assert torch.equal(outs[0]["logits"], outs[1]["logits"])
    assert all(o["n_local"] > o["n_own"] for o in outs) and all(sum(o["send"]) < 0.2 * n for o in outs)   # banded: thin halo


def _oracle_step(gr, x, n, sd, hidden, layers=8):
    from oracle.symgated_oracle import OracleModel, bce_loss
    om = OracleModel(2, 2, hidden, 16, layers, 64, "batch", dropout=0.0)
    om.load_state_dict(sd)
    om.train()
    want = om((gr["src"], gr["dst"], n), x, gr["e"])
    loss = bce_loss(want, gr["y"], gr["pos_weight"])
    loss.backward()
    return om, want.detach().squeeze(-1), loss.detach()


def _check_partitioned_step(outs, om, want, want_loss, rtol):
    want_g = {k: p.grad for k, p in om.named_parameters()}
    for o in outs:
        assert (torch.sigmoid(o["logits"]) - torch.sigmoid(want)).abs().max().item() < 1e-4
        assert abs(o["loss"].item() - want_loss.item()) < 1e-5
        check_grads(o["grads"], want_g, rtol=rtol)
        num = sum(((o["grads"][k] - want_g[k]).double() ** 2).sum().item() for k in want_g) ** 0.5
        den = sum((want_g[k].double() ** 2).sum().item() for k in want_g) ** 0.5
        assert num / den < 3e-3, f"relative L2 error of the full gradient {num / den:.2e}"
        for k, b in om.named_buffers():
            assert torch.allclose(o["buffers"][k].float(), b.float(), atol=1e-5, rtol=1e-4), k
    for k in outs[0]["grads"]:   # every rank holds the same bits: per-rank optimizers stay in step
        assert all(torch.equal(outs[0]["grads"][k], o["grads"][k]) for o in outs[1:]), k


def test_two_ranks_training_step_h256_matches_oracle_autograd(tmp_path):
    """BASELINE configs[4]'s combination: a destination-range partition, train mode, H = 256 - two ranks sharing the GPU
    against the oracle's autograd on the UNPARTITIONED graph (the single-rank twin is
    test_hip_training.py::test_training_step_matches_oracle_autograd[256]; same graph, same weights, same tolerances)."""
    from gnnome_amd.synth import make_graph
    from oracle.symgated_oracle import degree_features
    n, e, hidden = 3000, 30_000, 256
    gr = make_graph(n, e, seed=9)
    x = degree_features(gr["src"], gr["dst"], n)
    sd = random_state_dict(hidden, seed=3)
    om, want, want_loss = _oracle_step(gr, x, n, sd, hidden)
    case = dict(src=gr["src"], dst=gr["dst"], num_nodes=n, x=x, e=gr["e"], y=gr["y"], pos_weight=gr["pos_weight"], hidden=hidden, layers=8,
                state_dict=sd, train=True, device="cuda")
    outs = _run(2, case, tmp_path)
    assert sum(o["n_score"] for o in outs) == e and all(o["n_local"] > o["n_own"] for o in outs)
    _check_partitioned_step(outs, om, want, want_loss, rtol=3e-2)


@pytest.mark.parametrize("hidden", [128, 256])
def test_two_ranks_training_step_with_bf16_activation_storage(tmp_path, hidden):
    """Round 4: activation_storage = "bf16" on a partition (two ranks sharing the GPU, HIP kernels): against the SINGLE-rank bf16 step on the
    same GPU - same roundings of xe / dxe, the BatchNorm statistics summed across the cut in another order."""
    import torch.nn.functional as F
    import gnnome_amd
    from gnnome_amd.synth import make_graph
    from oracle.symgated_oracle import degree_features
    n, e = 3000, 30_000
    gr = make_graph(n, e, seed=9)
    x = degree_features(gr["src"], gr["dst"], n)
    sd = random_state_dict(hidden, seed=3)
    dev = torch.device("cuda", 0)
    m = gnnome_amd.models.SymGatedGCNModel(2, 2, hidden, 16, 8, 64, "batch", dropout=0.0)
    m.load_state_dict(sd)
    m.activation_storage = "bf16"
    m.to(dev).train()
    one = m((gr["src"].to(dev), gr["dst"].to(dev), n), x.to(dev), gr["e"].to(dev))
    loss1 = F.binary_cross_entropy_with_logits(one.squeeze(-1), gr["y"].to(dev), pos_weight=gr["pos_weight"].to(dev))
    loss1.backward()
    g1 = {k: p.grad.detach().cpu() for k, p in m.named_parameters()}
    case = dict(src=gr["src"], dst=gr["dst"], num_nodes=n, x=x, e=gr["e"], y=gr["y"], pos_weight=gr["pos_weight"], hidden=hidden, layers=8,
                state_dict=sd, train=True, device="cuda", activation_storage="bf16")
    outs = _run(2, case, tmp_path)
    assert sum(o["n_score"] for o in outs) == e and all(o["n_local"] > o["n_own"] for o in outs)
    for o in outs:
        assert abs(o["loss"].item() - loss1.item()) <= 1e-4 * abs(loss1.item())
        assert (torch.sigmoid(o["logits"]) - torch.sigmoid(one.detach().squeeze(-1).cpu())).abs().max().item() < 2e-3
        num = sum(((o["grads"][k] - g1[k]).double() ** 2).sum().item() for k in g1) ** 0.5
        den = sum((g1[k].double() ** 2).sum().item() for k in g1) ** 0.5
        assert num / den < 1e-2, num / den
    for k in outs[0]["grads"]:
        assert torch.equal(outs[0]["grads"][k], outs[1]["grads"][k]), k


def test_three_ranks_training_step_h128_on_a_mostly_cut_graph(tmp_path):
    """uniform graph at H = 128 on THREE ranks sharing the GPU: two thirds of the edges are cut (live on two ranks), so most gradient
    rows are assembled from partial sums (the fused BatchNorm-backward + data-gradient pass with rows_once < rows, halo gradients
    every way) - VERDICT r2 item 1(d)."""
    from gnnome_amd.synth import make_graph
    from oracle.symgated_oracle import degree_features
    n, e, hidden = 3000, 30_000, 128
    gr = make_graph(n, e, seed=11, kind="uniform")
    x = degree_features(gr["src"], gr["dst"], n)
    sd = random_state_dict(hidden, seed=5)
    om, want, want_loss = _oracle_step(gr, x, n, sd, hidden)
    case = dict(src=gr["src"], dst=gr["dst"], num_nodes=n, x=x, e=gr["e"], y=gr["y"], pos_weight=gr["pos_weight"], hidden=hidden, layers=8,
                state_dict=sd, train=True, device="cuda")
    outs = _run(3, case, tmp_path)
    assert sum(o["e_local"] for o in outs) > 1.6 * e       # > 60 % of the edges are cut: replicated on a second rank
    _check_partitioned_step(outs, om, want, want_loss, rtol=3e-2)


def test_rccl_collective_wrappers_on_device_memory(tmp_path):
    """gnnome_amd.dist's three wrappers over RCCL ("nccl") with device tensors: rows that really travel through
    all_to_all_single (async + wait, a rank sending to itself), all_reduce, all_gather_into_tensor (fp32 and int64)."""
    o = _run(1, dict(wrappers=True, device="cuda", transport="nccl"), tmp_path)[0]
    assert o["backend"] == "nccl"
    assert torch.equal(o["got"], o["sent"]) and torch.equal(o["summed"], o["sent"][:7]) and torch.equal(o["every"][0], o["sent"][:5])
    assert o["ints"].tolist() == [[0, 1, 2, 3]]


def test_one_rank_over_rccl_with_forced_collectives_equals_the_plain_path(tmp_path):
    """The first time RCCL runs gnnome_amd.dist: init_process_group("nccl") with one rank and
    GNNOME_FORCE_COLLECTIVES=1, so that the plan's all_to_all, every layer's halo exchange (async, overlapped with the owned
    rows' projection), the logits all_gather + index_select and, in training, the BatchNorm statistics all_gather, the
    BatchNorm-backward / halo-gradient / flat gradient all_reduce all go through RCCL on device memory.  Inference: the same
    bits as the plain single-rank partitioned path; training: the reference golden G3, and the H = 256 banded graph equal to
    the un-forced step to rounding of the per-channel statistics."""
    from gnnome_amd.synth import make_graph
    from oracle.symgated_oracle import degree_features
    n, e, hidden = 20_000, 200_000, 256
    gr = make_graph(n, e, seed=3, kind="banded")
    case = dict(src=gr["src"], dst=gr["dst"], num_nodes=n, x=degree_features(gr["src"], gr["dst"], n), e=gr["e"], hidden=hidden, layers=8,
                state_dict=random_state_dict(hidden, seed=2), device="cuda", captured=True)
    plain = _run(1, case, tmp_path)[0]
    rccl = _run(1, dict(case, transport="nccl", force_collectives=True, sliced=True), tmp_path)[0]   # (+ the plan from edge-list slices)
    assert rccl["backend"] == "nccl" and rccl["score_index"] and not plain["score_index"] and rccl["sliced_equal"]
    assert torch.equal(plain["logits"], rccl["logits"]) and torch.equal(rccl["replayed"], rccl["logits"])

    g = load_golden("g3_train_h64.pt")
    tcase = dict(src=g["src"], dst=g["dst"], num_nodes=g["num_nodes"], x=g["x"], e=g["e"], y=g["y"], pos_weight=g["pos_weight"],
                 hidden=64, layers=8, state_dict=random_state_dict(64, seed=g["seed"]), train=True, device="cuda", transport="nccl",
                 force_collectives=True)
    o = _run(1, tcase, tmp_path)[0]
    assert o["backend"] == "nccl" and abs(o["loss"].item() - g["loss"].item()) < 1e-5
    check_grads(o["grads"], g["grads"], rtol=1e-3)

    big = dict(case, captured=False, train=True, y=gr["y"], pos_weight=gr["pos_weight"])
    a = _run(1, big, tmp_path)[0]
    b = _run(1, dict(big, transport="nccl", force_collectives=True), tmp_path)[0]
    assert abs(a["loss"].item() - b["loss"].item()) < 1e-6 * abs(a["loss"].item()) + 1e-7
    check_grads(b["grads"], a["grads"], rtol=2e-3)
