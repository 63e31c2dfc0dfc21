"""Greedy decode (inference.py:29-67, 70-165, 167-359): the oracle against walks produced by the
reference's own functions (tests/golden/g9_decode.pt, made by make_golden_decode.py from inference.py's syntax tree), and
the HIP path (gnnome_amd/decode.py, csrc/decode.hip) against the oracle - integer walks and contig lengths bit-exact."""
import os
import sys

import pytest
import torch

from conftest import GOLDEN, load_golden
from oracle import decode_oracle as oracle

sys.path.insert(0, GOLDEN)


def _cases():
    return load_golden("g9_decode.pt")["cases"]


def dev():
    return torch.device("cuda", 0)


# ------------------------------------------------------------------------------------------------ CPU: oracle vs reference

def _nth0_libstdcxx(vals):
    """csrc/decode.hip's nth0_libstdcxx, in Python: std::nth_element(first, first, last, greater) of libstdc++."""
    v, p, n = list(vals), list(range(len(vals))), len(vals)

    def swp(a, b):
        v[a], v[b] = v[b], v[a]
        p[a], p[b] = p[b], p[a]
    first, last, depth, m = 0, n, 0, n
    while m > 1:
        depth, m = depth + 1, m >> 1
    depth *= 2
    while last - first > 3:
        if depth == 0:
            for i in range(first + 1, last):
                if v[i] > v[first]:
                    swp(i, first)
            return p[0]
        depth -= 1
        a, b, c = first + 1, first + (last - first) // 2, last - 1
        if v[a] > v[b]:
            swp(first, b if v[b] > v[c] else (c if v[a] > v[c] else a))
        else:
            swp(first, a if v[a] > v[c] else (c if v[b] > v[c] else b))
        lo, hi = first + 1, last
        while True:
            while v[lo] > v[first]:
                lo += 1
            hi -= 1
            while v[first] > v[hi]:
                hi -= 1
            if not lo < hi:
                break
            swp(lo, hi)
            lo += 1
        last = lo
    for i in range(first + 1, last):
        tv, tp = v[i], p[i]
        j = i
        if tv > v[first]:
            while j > first:
                v[j], p[j] = v[j - 1], p[j - 1]
                j -= 1
        else:
            while tv > v[j - 1]:
                v[j], p[j] = v[j - 1], p[j - 1]
                j -= 1
        v[j], p[j] = tv, tp
    return p[0]


def test_tie_rule_of_torch_topk_is_the_one_the_kernel_replays():
    """Which of several EQUAL maxima torch.topk(k=1) returns on the CPU (saturated scores give log-probabilities of exactly
    0): libstdc++'s nth_element below 64 candidates - not the first maximum - and the first maximum from 64 up."""
    import random
    assert int(torch.topk(torch.tensor([0., 0., -.2, 0., 0., -.7, 0.]), k=1, dim=0)[1]) == 3
    rnd = random.Random(0)
    for _ in range(4000):
        vals = [rnd.choice([0.0, 0.0, -0.2, -0.7, -1.5, 0.0]) for _ in range(rnd.randint(2, 63))]
        assert int(torch.topk(torch.tensor(vals), k=1, dim=0)[1]) == _nth0_libstdcxx(vals), vals
    for n in (64, 65, 100, 700):
        for _ in range(50):
            t = torch.tensor([rnd.choice([0.0, -0.2, 0.0, -3.0]) for _ in range(n)])
            assert int(torch.topk(t, k=1, dim=0)[1]) == int((t == t.max()).nonzero()[0])


def test_oracle_walks_equal_the_reference_functions():
    for c in _cases():
        succs, preds, edges = oracle.neighbor_dicts(c["src"], c["dst"], c["num_nodes"])
        logp = torch.log(torch.sigmoid(c["scores"]))
        visited = set(c["visited"])
        for cand in c["candidates"]:
            s, d = int(c["src"][cand["edge"]]), int(c["dst"][cand["edge"]])
            walk_f, walk_b, visited_f, visited_b, sum_f, sum_b = oracle.run_greedy_both_ways(s, d, logp, succs, preds, edges, visited)
            assert walk_f == cand["walk_f"] and walk_b == cand["walk_b"], (c["name"], cand["edge"])
            assert float(sum_f) == cand["sum_f"] and float(sum_b) == cand["sum_b"]
            assert visited_f == set(walk_f) | {w ^ 1 for w in walk_f}


def test_oracle_sampling_equals_the_reference_function():
    for c in _cases():
        torch.manual_seed(c["sample_seed"])
        idx = oracle.sample_edges(torch.sigmoid(c["scores"][c["sample_remaining"]]), 24)
        assert torch.equal(idx, c["sampled_index"])


def test_oracle_outer_loop_consumes_the_graph():
    c = _cases()[0]
    torch.manual_seed(3)
    trace = []
    walks = oracle.get_contigs_greedy(c["src"], c["dst"], c["num_nodes"], c["scores"], c["prefix_length"], c["read_length"],
                                      len_threshold=20_000, nb_paths=12, trace=trace)
    assert len(walks) >= 3 and all(len(w) >= 1 for w in walks)
    seen = set()
    for w in walks:   # contigs never share a read (either strand)
        reads = {x >> 1 for x in w}
        assert not (reads & seen)
        seen |= reads
    assert trace[-1]["contig_len"] < 20_000 or len(seen) * 2 >= c["num_nodes"] - 2


# ------------------------------------------------------------------------------------------------ GPU: HIP vs oracle

def _decode_graph(c, on_device=False):
    from gnnome_amd import decode
    dg = decode.DecodeGraph(c["src"], c["dst"], c["num_nodes"], c["prefix_length"], c["read_length"], device=dev())
    return dg.set_scores(c["scores"], logprobs_on_device=on_device)


@pytest.mark.gpu
def test_hip_walks_equal_the_reference_goldens():
    from gnnome_amd import decode
    for c in _cases():
        dg = _decode_graph(c)
        visited = torch.zeros(c["num_nodes"], dtype=torch.uint8, device=dev())
        if c["visited"]:
            visited[torch.tensor(c["visited"], device=dev())] = 1
        cand = torch.tensor([k["edge"] for k in c["candidates"]])
        res = decode.greedy_walks(dg, visited, cand)
        assert int(res.status.max()) == 0
        succs, preds, edges = oracle.neighbor_dicts(c["src"], c["dst"], c["num_nodes"])
        for i, k in enumerate(c["candidates"]):
            lf, lb = int(res.len_f[i]), int(res.len_b[i])
            assert res.walks_f[i, :lf].tolist() == k["walk_f"], (c["name"], i)
            assert (torch.flip(res.walks_b[i, :lb], [0]) ^ 1).tolist() == k["walk_b"], (c["name"], i)
            assert res.contig(i) == k["walk_b"] + k["walk_f"]
            # fp32 sums in walk order (only printed by the reference); the golden's log(sigmoid()) comes from another host's libm
            assert abs(float(res.sum_f[i]) - k["sum_f"]) <= 2e-6 * max(1.0, abs(k["sum_f"])) and abs(float(res.sum_b[i]) - k["sum_b"]) <= 2e-6 * max(1.0, abs(k["sum_b"]))
            want = oracle.contig_length(k["walk_b"] + k["walk_f"], edges, c["prefix_length"], c["read_length"])
            assert int(res.contig_len[i]) == want


@pytest.mark.gpu
@pytest.mark.parametrize("case", [0, 1, 3])
def test_hip_outer_loop_equals_the_oracle(case):
    from gnnome_amd import decode
    c = _cases()[case]
    torch.manual_seed(5)
    want_trace = []
    want = oracle.get_contigs_greedy(c["src"], c["dst"], c["num_nodes"], c["scores"], c["prefix_length"], c["read_length"],
                                     len_threshold=15_000, nb_paths=10, trace=want_trace)
    torch.manual_seed(5)
    stats = []
    got = decode.decode_contigs(_decode_graph(c), 15_000, nb_paths=10, stats=stats)
    assert got == want
    assert [s["contig_len"] for s in stats] == [t["contig_len"] for t in want_trace]


@pytest.mark.gpu
def test_hip_get_contigs_greedy_signature_and_checkpoint(tmp_path):
    """The reference's call (inference.py:167): a graph object + the three dicts, which are not read."""
    from gnnome_amd import decode
    c = _cases()[2]

    class G:
        ndata = {"read_length": c["read_length"]}
        edata = {"score": c["scores"].unsqueeze(1), "prefix_length": c["prefix_length"]}

        def edges(self):
            return c["src"], c["dst"]

        def num_nodes(self):
            return c["num_nodes"]

    torch.manual_seed(9)
    want = oracle.get_contigs_greedy(c["src"], c["dst"], c["num_nodes"], c["scores"], c["prefix_length"], c["read_length"], 10_000, nb_paths=8)
    torch.manual_seed(9)
    got = decode.get_contigs_greedy(G(), None, None, None, 10_000, nb_paths=8, checkpoint_dir=str(tmp_path))
    assert got == want
    if len(got) >= 10:
        assert os.path.isfile(tmp_path / "checkpoint.pkl")


@pytest.mark.gpu
def test_hip_parallel_edges_capacity_and_asymmetric_graphs():
    from gnnome_amd import decode
    # parallel edges: the reference's edges dict keeps the LAST id of a pair; both slots must rank with that id's score
    src = torch.tensor([0, 0, 0, 2, 4, 3, 5, 3, 1])
    dst = torch.tensor([2, 2, 4, 6, 6, 1, 1, 1, 7])     # (0,2) twice, (3,1) twice; mates present for the walked edges
    n = 8
    scores = torch.tensor([5.0, -3.0, 0.0, 1.0, 1.0, 0.5, 0.2, 0.1, 0.3])
    prefix = torch.arange(1, 10) * 100
    rl = torch.full((n,), 1000)
    dg = decode.DecodeGraph(src, dst, n, prefix, rl, device=dev()).set_scores(scores)
    succs, preds, edges = oracle.neighbor_dicts(src, dst, n)
    assert edges[(0, 2)] == 1
    logp = torch.log(torch.sigmoid(scores))
    visited = torch.zeros(n, dtype=torch.uint8, device=dev())
    res = decode.greedy_walks(dg, visited, torch.tensor([8]))     # start edge (1, 7): forward from 7 (dead end), backward from 0
    wf, wb, *_ = oracle.run_greedy_both_ways(1, 7, logp, succs, preds, edges, set())
    assert res.contig(0) == wb + wf
    # capacity: a cut walk is reported, not silently truncated
    c = _cases()[0]
    dgc = _decode_graph(c)
    res = decode.greedy_walks(dgc, torch.zeros(c["num_nodes"], dtype=torch.uint8, device=dev()), torch.tensor([c["candidates"][5]["edge"]]), capacity=4)
    assert int(res.status[0]) & 1


@pytest.mark.gpu
def test_hip_decode_large_graph_properties():
    """100k-read graph, 100 candidates per iteration (hyperparameters.py:47): contigs are vertex-disjoint over reads, every
    step follows an edge, and two runs from the same seed give the same walks."""
    import time

    from make_golden_decode import decode_graph

    from gnnome_amd import decode
    g = decode_graph(100_000, 4, seed=21, saturate=0.05)
    dg = decode.DecodeGraph(g["src"], g["dst"], g["num_nodes"], g["prefix_length"], g["read_length"], device=dev()).set_scores(g["scores"])
    runs = []
    for _ in range(2):
        torch.manual_seed(1)
        stats = []
        t0 = time.perf_counter()
        runs.append(decode.decode_contigs(dg, 200_000, nb_paths=100, stats=stats))
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"decode: N={g['num_nodes']} E={g['src'].numel()}: {len(runs[-1])} contigs, {sum(s['walk_len'] for s in stats)} nodes in the chosen "
              f"walks, {len(stats)} iterations x 100 candidates in {dt:.2f} s")
    assert runs[0] == runs[1] and len(runs[0]) >= 5
    pairs = set(zip(g["src"].tolist(), g["dst"].tolist()))
    seen = set()
    for w in runs[0]:
        assert all((a, b) in pairs for a, b in zip(w[:-1], w[1:]))
        reads = {x >> 1 for x in w}
        assert len(reads) == len(w) and not (reads & seen)
        seen |= reads
