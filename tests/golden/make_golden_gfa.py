"""Generate tests/golden/g10_gfa.pt (+ g10_*.gfa inputs) from the REFERENCE's own GFA parser.  Build container only:

    python tests/golden/make_golden_gfa.py        # needs /root/reference (read-only)

graph_parser.py imports Bio, dgl and edlib, none of which exist in this image, so `only_from_gfa` (:120-581) and the
three dict builders (:18-80) are compiled from the file's syntax tree and run with
  * networkx - the real one (3.4.2 here) - for the graph the parser builds,
  * `Seq`: a str subclass with reverse_complement() (all the parser uses of Biopython on this path),
  * `dgl.from_networkx`: edges and attributes read off the networkx graph in ITS OWN iteration order after
    nx.convert_node_labels_to_integers(ordering='sorted') - what DGL 0.8.1 documents; DGL cannot be run here, so the edge
    NUMBERING is pinned through networkx's order only,
  * `edlib.align`: the global (NW) edit distance as a plain dynamic programme - its documented result.
The reference's text runs; none of it is stored.  Inputs are three small synthetic GFAs in the three L-line dialects the
parser accepts (raven / GFA 1: 6 fields, hifiasm: 7, newer hifiasm: 8), with all four orientation cases, a repeated
link, a zero-length overlap and a unitig segment followed by A lines."""
import ast
import gzip
import os
import re
import sys
from collections import Counter, namedtuple
from datetime import datetime

import networkx as nx
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"

_COMP = str.maketrans("ACGTacgtNn", "TGCAtgcaNn")


class Seq(str):
    def reverse_complement(self):
        return Seq(self.translate(_COMP)[::-1])


def edit_distance(a, b):
    prev = list(range(len(b) + 1))
    for i, ca in enumerate(a, 1):
        cur = [i]
        for j, cb in enumerate(b, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (ca != cb)))
        prev = cur
    return prev[-1]


class _Edlib:
    @staticmethod
    def align(a, b):
        return {"editDistance": edit_distance(a, b)}


class _Graph:
    def __init__(self, src, dst, n, ndata, edata):
        self._s, self._d, self._n, self.ndata, self.edata = src, dst, n, ndata, edata

    def nodes(self):
        return torch.arange(self._n)

    def edges(self):
        return self._s, self._d

    def num_nodes(self):
        return self._n


class _Dgl:
    @staticmethod
    def from_networkx(nx_graph, node_attrs=None, edge_attrs=None):
        g = nx.convert_node_labels_to_integers(nx_graph, ordering="sorted")
        pairs = list(g.edges)
        src = torch.tensor([u for u, _ in pairs], dtype=torch.int64)
        dst = torch.tensor([v for _, v in pairs], dtype=torch.int64)
        ndata = {a: torch.tensor([g.nodes[i][a] for i in range(g.number_of_nodes())]) for a in (node_attrs or [])}
        edata = {a: torch.tensor([g.edges[u, v][a] for u, v in pairs]) for a in (edge_attrs or [])}
        return _Graph(src, dst, g.number_of_nodes(), ndata, edata)


def reference_functions(path, names, scope):
    tree = ast.parse(open(path).read())
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in names:
            exec(compile(ast.Module([node], []), path, "exec"), scope)
    return [scope[n] for n in names]


def synthetic_gfa(path, dialect, reads, seed, with_unitig=False):
    rng = np.random.default_rng(seed)
    seqs = ["".join(rng.choice(list("ACGT"), size=int(rng.integers(120, 260)))) for _ in range(reads)]
    ids = [f"read{k}" for k in range(reads)]
    if with_unitig:
        ids[3] = "utg000004l"
    with open(path, "w") as f:
        for k, (rid, s) in enumerate(zip(ids, seqs)):
            f.write(f"S\t{rid}\t{s}\tLN:i:{len(s)}\n")
            if rid.startswith("utg"):
                f.write(f"A\t{rid}\t0\t+\tm54_{k}/1/ccs\t0\t{len(s) // 2}\tid:i:1\n")
                f.write(f"A\t{rid}\t{len(s) // 2}\t-\tm54_{k}/2/ccs\t0\t{len(s) // 2}\tid:i:2\n")
        links = []
        for k in range(reads):
            for off in range(1, 1 + int(rng.integers(1, 4))):
                t = k + off
                if t < reads:
                    links.append((k, "+-"[int(rng.integers(0, 2))], t, "+-"[int(rng.integers(0, 2))], int(rng.integers(20, 100))))
        links.append((links[2][0], links[2][1], links[2][2], links[2][3], links[2][4] + 7))   # repeated link: last attributes win
        links.append((1, "+", 5, "+", 0))                                                       # zero-length overlap: skipped
        order = rng.permutation(len(links))
        for j in order:
            a, oa, b, ob, ol = links[j]
            ia, ib = ids[a], ids[b]
            if dialect == 6:
                f.write(f"L\t{ia}\t{oa}\t{ib}\t{ob}\t{ol}M\n")
            elif dialect == 7:
                f.write(f"L\t{ia}:1-{len(seqs[a])}\t{oa}\t{ib}:1-{len(seqs[b])}\t{ob}\t{ol}M\tL1:i:{len(seqs[a]) - ol}\n")
            else:
                f.write(f"L\t{ia}\t{oa}\t{ib}\t{ob}\t{ol}M\tL1:i:{len(seqs[a]) - ol}\tL2:i:{len(seqs[b]) - ol}\n")


def main():
    scope = {"nx": nx, "Seq": Seq, "dgl": _Dgl, "edlib": _Edlib, "tqdm": lambda x, **k: x, "datetime": datetime, "re": re, "gzip": gzip,
             "Counter": Counter, "namedtuple": namedtuple, "SeqIO": None, "print": lambda *a, **k: None}
    (only_from_gfa,) = reference_functions(os.path.join(REF, "graph_parser.py"),
                                           ["get_neighbors", "get_predecessors", "get_edges", "calculate_similarities", "only_from_gfa"], scope)[-1:]
    cases = []
    for name, dialect, reads, seed, utg in [("raven6", 6, 40, 1, False), ("hifiasm7", 7, 30, 2, False), ("hifiasm8_utg", 8, 30, 3, True)]:
        path = os.path.join(HERE, f"g10_{name}.gfa")
        synthetic_gfa(path, dialect, reads, seed, utg)
        g, aux = only_from_gfa(path, training=False, reads_path=None, get_similarities=True)
        src, dst = g.edges()
        cases.append({"name": name, "gfa": os.path.basename(path), "src": src, "dst": dst, "num_nodes": g.num_nodes(),
                      "read_length": g.ndata["read_length"], "prefix_length": g.edata["prefix_length"],
                      "overlap_length": g.edata["overlap_length"], "overlap_similarity": g.edata["overlap_similarity"],
                      "succ": aux["succ"], "pred": aux["pred"], "edges": aux["edges"], "read_to_node": aux["read_to_node"],
                      "node_to_read": aux["node_to_read"]})
        print(name, "N =", g.num_nodes(), "E =", src.numel())
    torch.save({"cases": cases, "made_with": "tests/golden/make_golden_gfa.py (graph_parser.only_from_gfa via ast, networkx " + nx.__version__ + ")"},
               os.path.join(HERE, "g10_gfa.pt"))


if __name__ == "__main__":
    main()
