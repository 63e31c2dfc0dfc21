"""Host-side mirrors of the reference's layer classes for the SymGatedGCN path.

Same class names, constructor arguments, parameter / buffer names and forward signatures as
layers/gated_gcn_full.py:8-42,82-142 (SymGatedGCN), layers/processor.py:9-19
(SymGatedGCN_processor) and layers/score_predictor.py:5-24 (ScorePredictor), so a reference
state_dict loads unchanged.  The modules only OWN parameters; all arithmetic is done by the HIP
kernels behind include/gnnome_hip.h (see engine.py).  The layer-level forwards accept and return
edge tensors in DGL edge-id order like the reference; the model-level forward keeps them in
destination-sorted order across the whole stack and never permutes an [E,H] tensor.
"""
import torch
import torch.nn as nn

from . import engine
from .graph import views_for


class SymGatedGCN(nn.Module):
    arithmetic = "auto"   # see SymGatedGCNModel.arithmetic (layer-level API)

    def __init__(self, in_channels, out_channels, normalization, dropout=None, residual=True):
        super().__init__()
        if in_channels != out_channels:
            raise ValueError("the SymGatedGCN path is only ever built with in_channels == out_channels "
                             "(layers/processor.py:12-14); unequal widths are not supported")
        if not residual:
            raise ValueError("residual=False is never used by the reference drivers and is not supported")
        # built widths: 64 (the reference's default, configs/hyperparameters.py:22), 128, 256; other widths up to 256 run zero-padded on the
        # next built one (engine.BUILT_HIDDEN; train mode: train._padded_step) - with LayerNorm the statistics run over the model's own channels
        engine.padded_width(in_channels)   # (raises above the largest built width)
        self.dropout = dropout if dropout else 0.0
        self.normalization = normalization
        self.residual = residual
        dtype = torch.float32
        self.A_1 = nn.Linear(in_channels, out_channels, dtype=dtype)
        self.A_2 = nn.Linear(in_channels, out_channels, dtype=dtype)
        self.A_3 = nn.Linear(in_channels, out_channels, dtype=dtype)
        self.B_1 = nn.Linear(in_channels, out_channels, dtype=dtype)
        self.B_2 = nn.Linear(in_channels, out_channels, dtype=dtype)
        self.B_3 = nn.Linear(in_channels, out_channels, dtype=dtype)
        if normalization == "batch":
            self.bn_h = nn.BatchNorm1d(out_channels, track_running_stats=True)
            self.bn_e = nn.BatchNorm1d(out_channels, track_running_stats=True)
        elif normalization == "layer":
            self.bn_h = nn.LayerNorm(out_channels)
            self.bn_e = nn.LayerNorm(out_channels)
        else:
            # the reference calls self.bn_e unconditionally (gated_gcn_full.py:106), so 'none' raises
            # AttributeError there; refuse it up front instead.
            raise ValueError("normalization must be 'batch' or 'layer'")

    def forward(self, g, h, e):
        """(h[N,H], e[E,H] in edge-id order) -> (h', e') like gated_gcn_full.py:82-142."""
        return engine.layer_forward_edge_id_order(self, g, h, e)


class SymGatedGCN_processor(nn.Module):
    def __init__(self, num_layers, hidden_features, normalization, dropout=None):
        super().__init__()
        self.convs = nn.ModuleList([
            SymGatedGCN(hidden_features, hidden_features, normalization, dropout) for _ in range(num_layers)
        ])

    def forward(self, graph, h, e):
        for conv in self.convs:
            h, e = conv(graph, h, e)
        return h, e


class ScorePredictor(nn.Module):
    def __init__(self, in_features, hidden_edge_scores):
        super().__init__()
        # built: in_features in {64,128,256}, hidden_edge_scores in {32,64,128} (reference default 64, 64); anything below the largest
        # runs zero-padded on the next built width (engine.prepare_predictor), anything above is refused here
        engine.padded_width(in_features)
        engine.padded_width(hidden_edge_scores, engine.BUILT_SCORE_HIDDEN, "hidden_edge_scores")
        self.W1 = nn.Linear(3 * in_features, hidden_edge_scores)
        self.W2 = nn.Linear(hidden_edge_scores, 32)
        self.W3 = nn.Linear(32, 1)

    def forward(self, graph, x, e):
        """scores[E,1] in edge-id order from x[N,H], e[E,H] (edge-id order); score_predictor.py:19-24."""
        return engine.score_forward_edge_id_order(self, graph, x, e)


__all__ = ["SymGatedGCN", "SymGatedGCN_processor", "ScorePredictor", "views_for"]
