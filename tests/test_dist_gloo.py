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


@pytest.mark.parametrize("hidden", [128, 256])
def test_partitioned_training_step_with_bf16_activation_storage(hidden, tmp_path):
    """Round 4 (VERDICT r3 missing item 4, second half): activation_storage = "bf16" on a partition - xe / dxe of every rank's local
    edges as bfloat16, the BatchNorm statistics those of the ROUNDED owned rows, summed over ranks.  World 2 on the checker backend
    against the single-rank bf16 step of the same backend (same roundings, another summation order across the cut) and inside the
    bf16-vs-fp32 bounds of the single-rank test against the fp32 partitioned step."""
    import cpu_ops
    import gnnome_amd
    import torch.nn.functional as F
    from gnnome_amd import train as train_mod
    n, e, layers = 300, 3000, 2
    gr = make_graph(n, e, seed=8, kind="banded")
    x = degree_features(gr["src"], gr["dst"], n)
    sd = random_state_dict(hidden, num_layers=layers, seed=9)
    m = gnnome_amd.models.SymGatedGCNModel(2, 2, hidden, 16, layers, 64, "batch", dropout=0.0).train()
    m.load_state_dict(sd)
    m.activation_storage = "bf16"
    one = train_mod.train_forward_on(m, train_mod.WholeGraph(cpu_ops.CpuViews(gr["src"], gr["dst"], n), cpu_ops), x, gr["e"])
    loss1 = F.binary_cross_entropy_with_logits(one.squeeze(-1), gr["y"], pos_weight=gr["pos_weight"])
    loss1.backward()
    g1 = {k: p.grad.clone() for k, p in m.named_parameters()}
    case = dict(src=gr["src"], dst=gr["dst"], num_nodes=n, x=x, e=gr["e"], y=gr["y"], pos_weight=gr["pos_weight"], hidden=hidden, layers=layers,
                state_dict=sd, train=True)
    (tmp_path / "bf16").mkdir()
    (tmp_path / "fp32").mkdir()
    outs16 = _run(2, dict(case, activation_storage="bf16"), tmp_path / "bf16")
    outs32 = _run(2, case, tmp_path / "fp32")
    for o in outs16:
        assert abs(o["loss"].item() - loss1.item()) <= 2e-5 * abs(loss1.item())
        assert (torch.sigmoid(o["logits"]) - torch.sigmoid(one.detach().squeeze(-1))).abs().max().item() < 2e-3
        num = sum(((o["grads"][k] - g1[k]).double() ** 2).sum().item() for k in g1) ** 0.5
        den = sum((g1[k].double() ** 2).sum().item() for k in g1) ** 0.5
        assert num / den < 5e-3, num / den
        o32 = outs32[0]
        num = sum(((o["grads"][k] - o32["grads"][k]).double() ** 2).sum().item() for k in g1) ** 0.5
        den = sum((o32["grads"][k].double() ** 2).sum().item() for k in g1) ** 0.5
        assert abs(o["loss"].item() - o32["loss"].item()) < 2e-3 * abs(o32["loss"].item()) and num / den < 5e-2
    for k in outs16[0]["grads"]:
        assert torch.equal(outs16[0]["grads"][k], outs16[1]["grads"][k]), k


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


def test_one_rank_with_forced_collectives_equals_the_plain_path(tmp_path):
    """GNNOME_FORCE_COLLECTIVES=1 (gnnome_amd.dist.force_collectives): a one-rank group issues every collective of the
    partitioned path; a one-rank collective is the identity, so inference gives the same bits as without the switch and the
    training step the reference golden.  The GPU twin of this test runs the same code over RCCL (tests/test_hip_partition.py)."""
    from test_train_host import check_grads
    g = load_golden("g2_uniform_1k.pt")
    sd = random_state_dict(64, seed=3)
    case = dict(src=g["src"], dst=g["dst"], num_nodes=g["num_nodes"], x=g["x"], e=g["e"], hidden=64, layers=8, state_dict=sd)
    plain = _run(1, case, tmp_path)[0]
    forced = _run(1, dict(case, force_collectives=True), tmp_path)[0]
    assert forced["score_index"] and not plain["score_index"]          # the all-gather + index_select route was taken
    assert torch.equal(plain["logits"], forced["logits"])
    t = load_golden("g3_train_h64.pt")
    tcase = dict(src=t["src"], dst=t["dst"], num_nodes=t["num_nodes"], x=t["x"], e=t["e"], y=t["y"], pos_weight=t["pos_weight"],
                 hidden=64, layers=8, state_dict=random_state_dict(64, seed=t["seed"]), train=True, force_collectives=True)
    o = _run(1, tcase, tmp_path)[0]
    assert abs(o["loss"].item() - t["loss"].item()) < 1e-5
    check_grads(o["grads"], t["grads"], rtol=1e-3)


@pytest.mark.parametrize("world", [1, 2])
def test_collective_wrappers_move_rows(world, tmp_path):
    outs = _run(world, dict(wrappers=True), tmp_path)
    for r, o in enumerate(outs):
        for p in range(world):   # block p of what rank r received = block r of what rank p sent
            assert torch.equal(o["got"][p * 300:(p + 1) * 300], outs[p]["sent"][r * 300:(r + 1) * 300])
        assert torch.allclose(o["summed"], sum(q["sent"][:7] for q in outs))
        assert all(torch.equal(o["every"][p], outs[p]["sent"][:5]) for p in range(world))
        assert o["ints"].tolist() == [[p + k for k in range(4)] for p in range(world)]


def test_partitioned_training_step_with_recompute_gate(tmp_path):
    """model.recompute_gate on a partition (BASELINE configs[4]'s memory lever): two ranks, same bits as the stored form."""
    t = load_golden("g3_train_h64.pt")
    tcase = dict(src=t["src"], dst=t["dst"], num_nodes=t["num_nodes"], x=t["x"], e=t["e"], y=t["y"], pos_weight=t["pos_weight"],
                 hidden=64, layers=8, state_dict=random_state_dict(64, seed=t["seed"]), train=True)
    stored = _run(2, tcase, tmp_path)
    again = _run(2, dict(tcase, recompute_gate=True), tmp_path)
    for a, b in zip(stored, again):
        assert torch.equal(a["logits"], b["logits"]) and all(torch.equal(a["grads"][k], b["grads"][k]) for k in a["grads"])


@pytest.mark.parametrize("kind,world", [("banded", 2), ("uniform", 3)])
def test_partition_census_predicts_what_every_rank_builds(kind, world, tmp_path):
    """gnnome_amd.dist.partition_census (one process, no process group - tools/partition_stats.py and DESIGN.md's world-8
    tables rest on it) against the plans the ranks really build."""
    from gnnome_amd.dist import partition_census
    n, e = 3000, 30000
    gr = make_graph(n, e, seed=6, kind=kind)
    case = dict(src=gr["src"], dst=gr["dst"], num_nodes=n, x=degree_features(gr["src"], gr["dst"], n), e=gr["e"], hidden=64, layers=1,
                state_dict=random_state_dict(64, num_layers=1, seed=1))
    outs = _run(world, case, tmp_path)
    c = partition_census(gr["src"], gr["dst"], n, world)
    assert c["bounds"] == outs[0]["bounds"]
    for o, r in zip(outs, c["ranks"]):
        assert (o["n_own"], o["n_local"] - o["n_own"], o["e_local"], o["n_score"], sum(o["send"]), max(o["send"])) == \
            (r["owned_nodes"], r["halo_rows"], r["local_edges"], r["owned_in_edges"], r["rows_sent_per_layer"], r["largest_link_rows"])
    assert abs(c["edge_replication"] - sum(o["e_local"] for o in outs) / e) < 1e-12
    assert (c["cut_fraction"] < 0.05) == (kind == "banded")


@pytest.mark.parametrize("kind,world", [("banded", 2), ("uniform", 3), ("banded", 1)])
def test_plan_from_edge_list_slices_equals_plan_from_the_whole_list(kind, world, tmp_path):
    """PartitionedGraph.from_slices: every rank starts from ITS slice of the edge list (uneven slices), degrees are all-reduced,
    edges travel to the owners of their endpoints - and the plan, the views, the shuffled edge features and the z-scored degree
    features equal from_global's, so the forward gives the same logits (no rank needs the whole edge list)."""
    n, e = 3000, 30000
    gr = make_graph(n, e, seed=9, kind=kind)
    sd = random_state_dict(64, num_layers=2, seed=1)
    x = degree_features(gr["src"], gr["dst"], n)
    case = dict(src=gr["src"], dst=gr["dst"], num_nodes=n, x=x, e=gr["e"], hidden=64, layers=2, state_dict=sd)
    whole = _run(world, case, tmp_path)
    sliced = _run(world, dict(case, sliced=True), tmp_path)
    assert all(o["sliced_equal"] for o in sliced)
    assert all(torch.equal(a["logits"], b["logits"]) for a, b in zip(whole, sliced))


def test_partition_over_renumbered_nodes_gives_the_same_logits_and_a_banded_cut(tmp_path):
    """node_perm (gnnome_amd/node_order.py on the GPU; here: the inverse of the shuffle that made the ids useless): the ranges
    are cut in the renumbered order, x / logits stay in the caller's numbering.  A banded graph with shuffled read ids is cut
    like a uniform one; with the permutation that undoes the shuffle it is cut like the banded one, and both routes (whole edge
    list, slices) give the logits of the unpartitioned oracle."""
    import numpy as np
    from gnnome_amd.dist import partition_census
    n, e = 3000, 30000
    gr = make_graph(n, e, seed=6, kind="permuted")
    rp = np.random.default_rng(6 + 7919).permutation(n // 2)             # synth.make_graph's shuffle: read r -> rp[r]
    inv = np.empty(n // 2, dtype=np.int64)
    inv[rp] = np.arange(n // 2)
    perm = torch.from_numpy(np.stack([2 * inv, 2 * inv + 1], 1).reshape(-1))    # node 2 r' + s -> 2 inv[r'] + s
    assert partition_census(gr["src"], gr["dst"], n, 2)["cut_fraction"] > 0.4
    assert partition_census(perm[gr["src"].long()], perm[gr["dst"].long()], n, 2)["cut_fraction"] < 0.05
    sd = random_state_dict(64, num_layers=2, seed=1)
    x = degree_features(gr["src"], gr["dst"], n)
    case = dict(src=gr["src"], dst=gr["dst"], num_nodes=n, x=x, e=gr["e"], hidden=64, layers=2, state_dict=sd, node_perm=perm, sliced=True)
    outs = _run(2, case, tmp_path)
    with torch.no_grad():
        want = model_from_state_dict(sd).eval()((gr["src"], gr["dst"], n), x, gr["e"]).squeeze(1)
    for o in outs:
        assert o["sliced_equal"] and (torch.sigmoid(o["logits"]) - torch.sigmoid(want)).abs().max().item() < 1e-4
        assert o["n_local"] - o["n_own"] < 0.1 * n and o["e_local"] < 0.55 * e       # banded halo, banded replication


def test_partitioned_forward_and_training_step_at_a_width_between_the_built_ones(tmp_path):
    """hidden_features = 96 / hidden_edge_scores = 48 (the reference takes any width; engine.BUILT_HIDDEN, train._padded_step) over a two-rank
    destination-range partition on the checker backend: eval logits against the oracle, and the train-mode step - loss, gradients in the
    model's own shapes, identical on both ranks - against the oracle's autograd."""
    from oracle.symgated_oracle import OracleModel, bce_loss
    n, e, layers, hidden, hs = 300, 3000, 2, 96, 48
    gr = make_graph(n, e, seed=8, kind="banded")
    x = degree_features(gr["src"], gr["dst"], n)
    sd = random_state_dict(hidden, num_layers=layers, hidden_edge_scores=hs, seed=9)
    om = OracleModel(2, 2, hidden, 16, layers, hs, "batch", dropout=0.0)
    om.load_state_dict(sd)
    om.eval()
    with torch.no_grad():
        want_eval = om((gr["src"], gr["dst"], n), x, gr["e"]).squeeze(-1)
    om.train()
    want = om((gr["src"], gr["dst"], n), x, gr["e"])
    want_loss = bce_loss(want, gr["y"], gr["pos_weight"])
    want_loss.backward()
    g_want = {k: p.grad for k, p in om.named_parameters()}
    case = dict(src=gr["src"], dst=gr["dst"], num_nodes=n, x=x, e=gr["e"], y=gr["y"], pos_weight=gr["pos_weight"], hidden=hidden, hs=hs, layers=layers,
                state_dict=sd)
    (tmp_path / "eval").mkdir()
    (tmp_path / "train").mkdir()
    for o in _run(2, case, tmp_path / "eval"):
        assert (torch.sigmoid(o["logits"]) - torch.sigmoid(want_eval)).abs().max().item() < 1e-4
    outs = _run(2, dict(case, train=True), tmp_path / "train")
    for o in outs:
        assert abs(o["loss"].item() - want_loss.item()) <= 2e-5 * abs(want_loss.item())
        assert set(o["grads"]) == set(g_want)
        for k, w in g_want.items():
            assert o["grads"][k].shape == w.shape
            assert (o["grads"][k] - w).abs().max().item() <= 2e-3 * w.abs().max().item() + 2e-6, k
    for k in g_want:
        assert torch.equal(outs[0]["grads"][k], outs[1]["grads"][k]), k


# ---- eight ranks (round 5): the process-group size of BASELINE configs[3] / [4], end to end on the CPU -------------

@pytest.mark.parametrize("kind", ["banded", "uniform"])
def test_eight_ranks_inference_matches_the_oracle(kind, tmp_path):
    """World 8 on a ~2k-node graph.  banded: a rank's halo comes from its two neighbours plus a few long-range rows, some send lists are
    EMPTY (all_to_all with zero-row blocks, all-gather padding of ranks with different row counts); uniform: 7/8 of the edges are cut
    and every rank talks to every other.  Logits of every rank against the unpartitioned oracle; every edge scored exactly once."""
    n, e = 2048, 20480
    gr = make_graph(n, e, seed=11, kind=kind)
    sd = random_state_dict(64, num_layers=3, seed=5)
    x = degree_features(gr["src"], gr["dst"], n)
    case = dict(src=gr["src"], dst=gr["dst"], num_nodes=n, x=x, e=gr["e"], hidden=64, layers=3, state_dict=sd)
    outs = _run(8, case, tmp_path)
    with torch.no_grad():
        want = model_from_state_dict(sd).eval()((gr["src"], gr["dst"], n), x, gr["e"]).squeeze(1)
    for o in outs:
        assert o["logits"].shape == want.shape and (torch.sigmoid(o["logits"]) - torch.sigmoid(want)).abs().max().item() < 1e-4
        assert torch.equal(o["logits"], outs[0]["logits"])
    assert sum(o["n_own"] for o in outs) == n and sum(o["n_score"] for o in outs) == e
    assert all(sum(o["recv"]) == o["n_local"] - o["n_own"] for o in outs)
    empty_links = sum(1 for o in outs for c in o["send"] if c == 0)
    if kind == "banded":
        # the 1 % long-range edges give most rank pairs a row or two; some links stay empty
        assert empty_links > 8
        assert sum(o["e_local"] for o in outs) < 1.25 * e
    else:
        assert empty_links == 8              # only the diagonal
        assert sum(o["e_local"] for o in outs) > 1.8 * e


@pytest.mark.parametrize("kind", ["banded", "uniform"])
def test_eight_ranks_training_step_matches_oracle_autograd(kind, tmp_path):
    """One train-mode step at world 8 (configs[4]'s group size): BatchNorm statistics all-reduced with every edge counted once, halo
    gradients returned through mostly-empty (banded) or all-to-all (uniform) links, parameter gradients summed over eight ranks -
    loss, all gradients and the BatchNorm buffers against the oracle's autograd on the unpartitioned graph; identical on every rank."""
    from oracle.symgated_oracle import OracleModel, bce_loss
    from test_train_host import check_grads
    n, e, layers = 1024, 10240, 2
    gr = make_graph(n, e, seed=12, kind=kind)
    x = degree_features(gr["src"], gr["dst"], n)
    sd = random_state_dict(64, num_layers=layers, seed=3)
    om = OracleModel(2, 2, 64, 16, layers, 64, "batch", dropout=0.0)
    om.load_state_dict(sd)
    om.train()
    want = om((gr["src"], gr["dst"], n), x, gr["e"])
    want_loss = bce_loss(want, gr["y"], gr["pos_weight"])
    want_loss.backward()
    case = dict(src=gr["src"], dst=gr["dst"], num_nodes=n, x=x, e=gr["e"], y=gr["y"], pos_weight=gr["pos_weight"], hidden=64, layers=layers,
                state_dict=sd, train=True)
    outs = _run(8, case, tmp_path)
    assert sum(o["n_score"] for o in outs) == e
    for o in outs:
        assert (torch.sigmoid(o["logits"]) - torch.sigmoid(want.detach().squeeze(-1))).abs().max().item() < 1e-4
        assert abs(o["loss"].item() - want_loss.item()) < 1e-5
        check_grads(o["grads"], {k: p.grad for k, p in om.named_parameters()}, rtol=2e-3)
        for k, b in om.named_buffers():
            assert torch.allclose(o["buffers"][k].float(), b.float(), atol=1e-5, rtol=1e-4), k
    for k in outs[0]["grads"]:
        assert all(torch.equal(outs[0]["grads"][k], o["grads"][k]) for o in outs[1:]), k


def test_eight_ranks_plan_from_slices_and_ranks_that_own_almost_nothing(tmp_path):
    """from_slices at world 8 (uneven slices, rank order) equals from_global array by array and gives the same logits; and a graph whose
    edges all sit on a few nodes, so that the balanced split leaves several ranks with node ranges that carry (almost) no edges:
    empty local edge lists, empty halos, zero-row score blocks must still assemble the full result."""
    n, e = 2048, 20480
    gr = make_graph(n, e, seed=13, kind="banded")
    sd = random_state_dict(64, num_layers=2, seed=1)
    x = degree_features(gr["src"], gr["dst"], n)
    case = dict(src=gr["src"], dst=gr["dst"], num_nodes=n, x=x, e=gr["e"], hidden=64, layers=2, state_dict=sd)
    (tmp_path / "whole").mkdir()
    (tmp_path / "sliced").mkdir()
    whole = _run(8, case, tmp_path / "whole")
    sliced = _run(8, dict(case, sliced=True), tmp_path / "sliced")
    assert all(o["sliced_equal"] for o in sliced)
    assert all(torch.equal(a["logits"], b["logits"]) for a, b in zip(whole, sliced))
    # 64 nodes, all 600 edges among the first 3 of them: at most three ranks can own a node with edges, the others get edgeless ranges
    g = torch.Generator().manual_seed(3)
    src, dst = torch.randint(0, 3, (600,), generator=g).int(), torch.randint(0, 3, (600,), generator=g).int()
    feat = torch.randn(600, 2, generator=g)
    xs = degree_features(src, dst, 64)
    case = dict(src=src, dst=dst, num_nodes=64, x=xs, e=feat, hidden=64, layers=2, state_dict=sd)
    (tmp_path / "skew").mkdir()
    outs = _run(8, case, tmp_path / "skew")
    with torch.no_grad():
        want = model_from_state_dict(sd).eval()((src, dst, 64), xs, feat).squeeze(1)
    assert sum(1 for o in outs if o["e_local"] == 0) >= 4 and sum(o["n_own"] for o in outs) == 64 and sum(o["n_score"] for o in outs) == 600
    for o in outs:
        assert (torch.sigmoid(o["logits"]) - torch.sigmoid(want)).abs().max().item() < 1e-4 and torch.equal(o["logits"], outs[0]["logits"])
