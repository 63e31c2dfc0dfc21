"""CPU oracle for GNNome's greedy decode (inference.py:29-67, 70-165, 167-359).  TEST INFRASTRUCTURE ONLY.

Nothing under `gnnome_amd/` may import this file; tests/ use it as the checker of gnnome_amd/decode.py.

A restatement with the reference's own data structures - successor / predecessor dicts of lists in edge-id order and
an `edges[(u, v)] -> id` dict (graph_parser.py:18-80) - and its own control flow, line for line where it matters:

  neighbor_dicts         graph_parser.py:31-37, :55-58, :77-80   (get_neighbors, get_predecessors, get_edges)
  greedy_forwards        inference.py:70-114
  greedy_backwards_rc    inference.py:117-158
  run_greedy_both_ways   inference.py:161-165
  contig_length          inference.py:29-36   (DGL's g.edges[u, v] lookup = the edges dict here)
  sample_edges           inference.py:54-67
  get_contigs_greedy     inference.py:193-359 (outer loop; DGL's node_subgraph = "edges whose endpoints are unvisited, in
                         edge-id order"; printing, timing and the ThreadPoolExecutor(1) dropped; checkpointing kept out)

Module constants of the reference that the walk depends on: RANDOM = False, early_stopping = False (inference.py:26-28).

Pinning: tests/golden/g9_decode.pt holds walks produced by the reference's own greedy_forwards / greedy_backwards_rc /
run_greedy_both_ways / sample_edges, compiled from inference.py's syntax tree by tests/golden/make_golden_decode.py
(the file imports DGL and cannot be imported whole); tests/test_decode.py checks this restatement against them.
"""
import torch


def neighbor_dicts(src, dst, num_nodes):
    succs = {i: [] for i in range(num_nodes)}
    preds = {i: [] for i in range(num_nodes)}
    edges = {}
    for idx, (s, d) in enumerate(zip(src.tolist(), dst.tolist())):
        succs[s].append(d)
        preds[d].append(s)
        edges[(s, d)] = idx
    return succs, preds, edges


def _walk(start, logProbs, neighbors, edges, visited_old):
    """The loop shared by greedy_forwards (:72-114) and greedy_backwards_rc (:119-156)."""
    current = start
    walk = []
    visited = set()
    sumLogProb = torch.tensor([0.0])
    while True:
        walk.append(current)
        visited.add(current)
        visited.add(current ^ 1)
        neighs_current = neighbors[current]
        if len(neighs_current) == 0:
            break
        if len(neighs_current) == 1:
            neighbor = neighs_current[0]
            if neighbor in visited_old or neighbor in visited:
                break
            sumLogProb += logProbs[edges[current, neighbor]]
            current = neighbor
            continue
        masked_neighbors = [n for n in neighs_current if not (n in visited_old or n in visited)]
        neighbor_edges = [edges[current, n] for n in masked_neighbors]
        if not neighbor_edges:
            break
        neighbor_p = logProbs[neighbor_edges]
        logProb, index = torch.topk(neighbor_p, k=1, dim=0)
        sumLogProb += logProb
        current = masked_neighbors[index]
    return walk, visited, sumLogProb


def greedy_forwards(start, logProbs, neighbors, predecessors, edges, visited_old):
    return _walk(start, logProbs, neighbors, edges, visited_old)


def greedy_backwards_rc(start, logProbs, predecessors, neighbors, edges, visited_old):
    walk, visited, sumLogProb = _walk(start ^ 1, logProbs, neighbors, edges, visited_old)
    return list(reversed([w ^ 1 for w in walk])), visited, sumLogProb


def run_greedy_both_ways(src, dst, logProbs, succs, preds, edges, visited):
    tmp_visited = visited | {src, src ^ 1, dst, dst ^ 1}
    walk_f, visited_f, sumLogProb_f = greedy_forwards(dst, logProbs, succs, preds, edges, tmp_visited)
    walk_b, visited_b, sumLogProb_b = greedy_backwards_rc(src, logProbs, preds, succs, edges, tmp_visited | visited_f)
    return walk_f, walk_b, visited_f, visited_b, sumLogProb_f, sumLogProb_b


def contig_length(walk, edges, prefix_length, read_length):
    total = sum(int(prefix_length[edges[(a, b)]]) for a, b in zip(walk[:-1], walk[1:]))
    return total + int(read_length[walk[-1]])


def sample_edges(prob_edges, nb_paths):
    if prob_edges.shape[0] > 2 ** 24:
        prob_edges = prob_edges[:2 ** 24]
    prob_edges = prob_edges.masked_fill(prob_edges < 1e-9, 1e-9)
    prob_edges = prob_edges / prob_edges.sum()
    prob_edges_nb_paths = prob_edges.repeat(nb_paths, 1)
    return torch.distributions.categorical.Categorical(prob_edges_nb_paths).sample()


def get_contigs_greedy(src, dst, num_nodes, scores, prefix_length, read_length, len_threshold, nb_paths=50, sampler=None,
                       trace=None):
    """-> list of walks.  `sampler(prob, k)`: indices into the remaining edges (default: sample_edges)."""
    sampler = sampler or sample_edges
    succs, preds, edges = neighbor_dicts(src, dst, num_nodes)
    logProbs = torch.log(torch.sigmoid(scores.float()))
    src_l, dst_l = src.tolist(), dst.tolist()
    all_contigs, visited = [], set()
    while True:
        remaining = [k for k in range(len(src_l)) if src_l[k] not in visited and dst_l[k] not in visited]
        if not remaining:
            break
        prob_edges = torch.sigmoid(scores.float()[remaining])
        idx_edges = sampler(prob_edges, nb_paths)
        results = {}
        for idx in torch.as_tensor(idx_edges).tolist():
            s, d = src_l[remaining[idx]], dst_l[remaining[idx]]
            results[(s, d)] = run_greedy_both_ways(s, d, logProbs, succs, preds, edges, visited)
        all_walks, all_visited_iter, all_contig_lens = [], [], []
        for k, (walk_f, walk_b, visited_f, visited_b, _, _) in results.items():
            walk_it = walk_b + walk_f
            len_contig_it = contig_length(walk_it, edges, prefix_length, read_length)
            if k[0] == k[1]:
                len_contig_it = 0
            all_walks.append(walk_it)
            all_visited_iter.append(visited_f | visited_b)
            all_contig_lens.append(len_contig_it)
        best = max(all_contig_lens)
        idxx = all_contig_lens.index(best)
        best_walk, best_visited = all_walks[idxx], all_visited_iter[idxx]
        trans = set()
        for ss, dd in zip(best_walk[:-1], best_walk[1:]):
            t1 = set(succs[ss]) & set(preds[dd])
            trans = trans | t1 | {t ^ 1 for t in t1}
        best_visited = best_visited | trans
        if trace is not None:
            trace.append({"contig_len": best, "walk_len": len(best_walk), "candidates": len(results)})
        if best < len_threshold:
            break
        all_contigs.append(best_walk)
        visited |= best_visited
    return all_contigs
