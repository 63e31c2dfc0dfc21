"""Input-side widening: the GFA reader against graphs produced by the reference's own parser
(tests/golden/g10_gfa.pt, made by make_golden_gfa.py from graph_parser.only_from_gfa), and the strand-wise node mask +
induced subgraph (train.py:91-100)."""
import os

import pytest
import torch

from conftest import GOLDEN, load_golden
from gnnome_amd import gfa


def _edit_distance(a, b):
    prev = list(range(len(b) + 1))
    for i, ca in enumerate(a, 1):
        cur = [i]
        for j, cb in enumerate(b, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (ca != cb)))
        prev = cur
    return prev[-1]


def _similarity(src_seq, dst_seq, ol):   # graph_parser.py:110-111 with a plain dynamic programme in edlib's place
    return 1 - _edit_distance(src_seq[-ol:], dst_seq[:ol]) / ol


def test_gfa_reader_equals_the_reference_parser():
    for c in load_golden("g10_gfa.pt")["cases"]:
        g = gfa.read_gfa(os.path.join(GOLDEN, c["gfa"]), similarity=_similarity)
        assert g["num_nodes"] == c["num_nodes"]
        assert torch.equal(g["src"], c["src"]) and torch.equal(g["dst"], c["dst"]), c["name"]        # DGL's edge numbering
        for key in ("read_length", "prefix_length", "overlap_length"):
            assert torch.equal(g[key], c[key].long()), (c["name"], key)
        assert torch.allclose(g["overlap_similarity"].double(), c["overlap_similarity"].double(), atol=1e-7)
        assert g["node_to_read"] == c["node_to_read"]
        want_r2n = c["read_to_node"]
        assert (g["read_to_node2"] if g["read_to_node2"] else g["read_to_node"]) == want_r2n     # graph_parser.py:412-413
        # the three dicts the decoder is handed (graph_parser.py:409-411) are functions of the edge list in this order
        from oracle.decode_oracle import neighbor_dicts
        succs, preds, edges = neighbor_dicts(g["src"], g["dst"], g["num_nodes"])
        assert succs == c["succ"] and preds == c["pred"] and edges == c["edges"]
        # strand symmetry the decoder relies on: every edge has its mate
        pairs = set(zip(g["src"].tolist(), g["dst"].tolist()))
        assert all((v ^ 1, u ^ 1) in pairs for u, v in pairs)


def test_gfa_similarity_sources(tmp_path):
    c = load_golden("g10_gfa.pt")["cases"][0]
    path = os.path.join(GOLDEN, c["gfa"])
    assert gfa.read_gfa(path, similarity=None)["overlap_similarity"] is None       # no aligner, no tags: None, not a guess
    sims = {(int(u), int(v)): float(s) for u, v, s in zip(c["src"], c["dst"], c["overlap_similarity"])}
    tagged = tmp_path / "tagged.gfa"
    gfa.write_similarity_tags(path, tagged, sims)
    g = gfa.read_gfa(tagged, similarity=None)
    assert torch.equal(g["src"], c["src"]) and torch.allclose(g["overlap_similarity"].double(), c["overlap_similarity"].double(), atol=1e-7)


def dev():
    return torch.device("cuda", 0)


@pytest.mark.gpu
def test_strandwise_mask_and_induced_subgraph():
    """train.py:91-100: both strands of a read are kept or dropped together; dgl.node_subgraph semantics."""
    import gnnome_amd
    from gnnome_amd import features
    from gnnome_amd.synth import make_graph, random_state_dict
    n, e = 4000, 40_000
    gr = make_graph(n, e, seed=4)
    torch.manual_seed(0)
    keep_half = torch.rand(n // 2) < 0.8                                           # the reference's draw (:92), CPU stream
    sub = features.mask_graph_strandwise((gr["src"], gr["dst"], n), 0.8, device=dev(), keep_half=keep_half)
    keep = torch.empty(n, dtype=torch.bool)
    keep[0::2], keep[1::2] = keep_half, keep_half                                  # :93-95
    nid = torch.nonzero(keep).squeeze(1)
    ek = keep[gr["src"].long()] & keep[gr["dst"].long()]
    eid = torch.nonzero(ek).squeeze(1)
    relabel = torch.full((n,), -1, dtype=torch.long)
    relabel[nid] = torch.arange(nid.numel())
    assert torch.equal(sub.nid.cpu(), nid) and torch.equal(sub.eid.cpu(), eid)
    assert torch.equal(sub.src.cpu().long(), relabel[gr["src"].long()[eid]]) and torch.equal(sub.dst.cpu().long(), relabel[gr["dst"].long()[eid]])
    assert sub.num_nodes() == int(keep.sum()) and sub.num_nodes() % 2 == 0
    # the subgraph goes through the model like any graph: same logits as a model call on the explicit edge list
    m = gnnome_amd.SymGatedGCNModel(2, 2, 64, 16, 2, 64, "batch").eval()
    m.load_state_dict({k: v for k, v in random_state_dict(64, num_layers=2, seed=2).items()})
    m.to(dev())
    x = features.degree_features(sub.views)
    ef = gr["e"].to(dev())[sub.eid]
    a = m(sub, x, ef)
    b = m((sub.src.cpu(), sub.dst.cpu(), sub.num_nodes()), x, ef)
    assert torch.equal(a, b) and a.shape == (eid.numel(), 1)
    # from prebuilt views of the whole graph, and a fresh draw on the device
    views = gnnome_amd.graph.views_for((gr["src"], gr["dst"], n), dev())
    sub2 = features.mask_graph_strandwise(views, 0.8, keep_half=keep_half)
    assert torch.equal(sub2.src, sub.src) and torch.equal(sub2.eid, sub.eid)
    sub3 = features.mask_graph_strandwise(views, 0.5)
    assert 0.3 * n < sub3.num_nodes() < 0.7 * n and bool((sub3.nid[0::2] + 1 == sub3.nid[1::2]).all())
