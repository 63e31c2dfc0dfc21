"""From a GFA to contig walks on one MI355X - the sequence of inference.py:411-467 with every step on this package's side
of the boundary (GFA reader, feature preparation, the SymGatedGCN scorer, greedy decode).  What inference.py does after it
(walks -> FASTA with the read sequences, evaluation with minigraph / paftools) is outside SURVEY.md 8's rows.

    model = gnnome_amd.SymGatedGCNModel(2, 2, 64, 16, 8, 64, 'batch'); model.load_state_dict(torch.load('weights/weights.pt')); model.eval()
    walks, scores, g = assemble('asm.gfa', model, len_threshold=10, nb_paths=100)
"""
import torch

from . import decode, features, gfa
from .graph import views_for


def score_graph(g, model, device=None):
    """inference.py:411-441: degree features (z-scored, :416-420), edge features (utils/data_utils.py:31-41), eval-mode model
    call -> logits[E] in edge-id order, on the device.  `g`: the dict read_gfa returns (overlap_similarity required - the
    shipped model was trained with it, hyperparameters.py:17)."""
    device = device or torch.device("cuda", torch.cuda.current_device())
    if g["overlap_similarity"] is None:
        raise ValueError("the graph carries no overlap similarities (see gnnome_amd.gfa: SI:f: tags, a similarity callable, or edlib)")
    views = views_for((g["src"], g["dst"], g["num_nodes"]), device)
    x = features.degree_features(views)
    e = features.edge_features(g["overlap_length"].to(device), g["overlap_similarity"].to(device))
    model = model.to(device).eval()
    with torch.no_grad():
        return model(views, x, e).squeeze(1)


def assemble(gfa_or_graph, model, len_threshold, nb_paths=100, similarity="auto", device=None, scores=None, sampler=None):
    """-> (walks, scores, graph dict).  `scores` overrides the model (inference.py:426-432: saved predictions / labels)."""
    device = device or torch.device("cuda", torch.cuda.current_device())
    g = gfa_or_graph if isinstance(gfa_or_graph, dict) else gfa.read_gfa(gfa_or_graph, similarity=similarity)
    if scores is None:
        scores = score_graph(g, model, device)
    prefix = g["prefix_length"].masked_fill(g["prefix_length"] < 0, 0)      # inference.py:461
    dg = decode.DecodeGraph(g["src"], g["dst"], g["num_nodes"], prefix, g["read_length"], device=device).set_scores(scores)
    return decode.decode_contigs(dg, len_threshold, nb_paths=nb_paths, sampler=sampler), scores, g
