"""Generate tests/golden/g9_decode.pt from the REFERENCE's own decode functions.  Build container only:

    python tests/golden/make_golden_decode.py        # needs /root/reference (read-only)

inference.py imports DGL, so its functions are compiled one at a time from the file's syntax tree (as
make_golden_closure.py does for train.py) and executed here with torch in scope - the reference's text runs, none of it
is stored.  Taken: greedy_forwards (:70-114), greedy_backwards_rc (:117-158), run_greedy_both_ways (:161-165),
sample_edges (:54-67), with the module constants they read (RANDOM, early_stopping, p_threshold: inference.py:26-28).
Inputs: small strand-symmetric synthetic graphs (gnnome_amd.synth conventions) with the successor / predecessor / edge
dicts built as graph_parser.py:31-37, :55-58, :77-80 build them.  The fixture is data only.
"""
import ast
import math
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(1, ROOT)


def reference_functions(path, names, scope):
    tree = ast.parse(open(path).read())
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in names:
            exec(compile(ast.Module([node], []), path, "exec"), scope)
    return [scope[n] for n in names]


def decode_graph(reads, mean_deg, seed, saturate=0.0):
    """A simple (no parallel edges) strand-symmetric overlap graph: read r overlaps the next few reads in layout order;
    every edge (u, v) has its mate (v^1, u^1).  prefix_length = read_length[src] - overlap_length (graph_parser.py:339-340)."""
    rng = np.random.default_rng(seed)
    pairs = set()
    for r in range(reads):
        for off in range(1, 1 + int(rng.integers(0, 2 * mean_deg + 1))):
            t = r + off
            if t < reads:
                s = int(rng.integers(0, 2))
                pairs.add((2 * r + s, 2 * t + s) if s == 0 else (2 * t + s, 2 * r + s))
    for _ in range(reads // 50):   # a few long-range ("repeat") edges
        a, b = int(rng.integers(0, reads)), int(rng.integers(0, reads))
        if a != b:
            pairs.add((2 * a, 2 * b))
    full = set()
    for u, v in pairs:
        full.add((u, v))
        full.add((v ^ 1, u ^ 1))
    edges = sorted(full)
    perm = rng.permutation(len(edges))
    src = torch.tensor([edges[i][0] for i in perm], dtype=torch.int64)
    dst = torch.tensor([edges[i][1] for i in perm], dtype=torch.int64)
    read_len = rng.integers(5000, 30000, size=reads)
    read_length = torch.from_numpy(np.repeat(read_len, 2).astype(np.int64))
    ol = {}
    for u, v in edges:
        canon = min((u, v), (v ^ 1, u ^ 1))
        if canon not in ol:
            ol[canon] = int(rng.integers(500, 4000))
    prefix_length = torch.tensor([int(read_length[u]) - ol[min((u, v), (v ^ 1, u ^ 1))] for u, v in zip(src.tolist(), dst.tolist())],
                                 dtype=torch.int64)
    scores = torch.from_numpy(rng.normal(0.0, 4.0, size=len(edges)).astype(np.float32))
    if saturate > 0:   # sigmoid == 1.0 exactly above ~17: log-probabilities tie at 0 and the FIRST successor must win
        hot = torch.from_numpy(rng.random(len(edges)) < saturate)
        scores[hot] = torch.from_numpy(rng.uniform(18.0, 40.0, size=int(hot.sum())).astype(np.float32))
    return {"src": src, "dst": dst, "num_nodes": 2 * reads, "scores": scores, "prefix_length": prefix_length, "read_length": read_length}


def dicts(src, dst, n):
    succs = {i: [] for i in range(n)}
    preds = {i: [] for i in range(n)}
    edges = {}
    for idx, (s, d) in enumerate(zip(src.tolist(), dst.tolist())):
        succs[s].append(d)
        preds[d].append(s)
        edges[(s, d)] = idx
    return succs, preds, edges


def main():
    scope = {"torch": torch, "math": math, "RANDOM": False, "early_stopping": False, "p_threshold": 0.06, "DEBUG": False}
    _, _, run_greedy_both_ways, sample_edges = reference_functions(
        os.path.join(REF, "inference.py"), ["greedy_forwards", "greedy_backwards_rc", "run_greedy_both_ways", "sample_edges"], scope)
    cases = []
    for name, reads, deg, seed, saturate, n_visited in [("plain", 300, 3, 11, 0.0, 0), ("ties", 200, 4, 12, 0.35, 0),
                                                         ("visited", 300, 3, 13, 0.1, 60), ("sparse", 150, 1, 14, 0.0, 10)]:
        g = decode_graph(reads, deg, seed, saturate)
        succs, preds, edges = dicts(g["src"], g["dst"], g["num_nodes"])
        logProbs = torch.log(torch.sigmoid(g["scores"]))     # inference.py:184
        gen = torch.Generator().manual_seed(seed)
        vis_reads = torch.randperm(reads, generator=gen)[:n_visited].tolist()
        visited = set()
        for r in vis_reads:
            visited |= {2 * r, 2 * r + 1}
        remaining = [k for k in range(g["src"].numel()) if int(g["src"][k]) not in visited and int(g["dst"][k]) not in visited]
        torch.manual_seed(seed)
        idx = sample_edges(torch.sigmoid(g["scores"][remaining]), 24)       # inference.py:211
        cand = [remaining[i] for i in idx.tolist()]
        out = []
        for k in cand:
            s, d = int(g["src"][k]), int(g["dst"][k])
            walk_f, walk_b, visited_f, visited_b, sum_f, sum_b = run_greedy_both_ways(s, d, logProbs, succs, preds, edges, visited)
            assert visited_f == set(walk_f) | {w ^ 1 for w in walk_f}
            out.append({"edge": k, "walk_f": walk_f, "walk_b": walk_b, "sum_f": float(sum_f), "sum_b": float(sum_b)})
        cases.append({"name": name, **g, "visited": sorted(visited), "sample_seed": seed, "sample_remaining": remaining,
                      "sampled_index": idx.clone(), "candidates": out})
        print(name, "E =", g["src"].numel(), "walk lengths", [len(c["walk_b"]) + len(c["walk_f"]) for c in out][:10])
    torch.save({"cases": cases, "made_with": "tests/golden/make_golden_decode.py (reference functions via ast)",
                "torch": torch.__version__}, os.path.join(HERE, "g9_decode.pt"))


if __name__ == "__main__":
    main()
