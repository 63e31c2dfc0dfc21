"""`SymGatedGCNModel` - drop-in for the reference's models/full_graph.py:9-30.

    model = SymGatedGCNModel(node_features, edge_features, hidden_features, hidden_ne_features,
                             num_layers, hidden_edge_scores, normalization, dropout=None)
    logits = model(graph, x, e)          # [E,1] fp32 logits in DGL edge-id order

Same positional constructor (inference.py:435, train.py:248), same 190 state_dict keys as
weights/weights.pt, same call.  Differences, all deliberate:
  * `graph` may be a DGLGraph, any object with .edges()/.num_nodes(), a (src, dst, N) tuple or a
    prebuilt gnnome_amd.ops.GraphViews.
  * inputs may live on the CPU (inference.py:388 pins device='cpu'): they are staged to the current
    HIP device and the logits are returned on the inputs' device.  The compute always runs on the
    MI355X; without the HIP library or a GPU the call raises.
  * the stray `print(x.shape)` of models/full_graph.py:25 is not reproduced.
  * `model.arithmetic` ("auto" | "reference" | "fast", default "auto") chooses, per layer, between the bf16x6
    matrix-core kernels and kernels that evaluate the layer's dense products in the reference's own ORDER (bit for bit
    what torch's CPU nn.Linear computes); "auto" uses the latter for layers whose eval-BatchNorm gain magnifies fp32
    reorder noise (engine.REFERENCE_ORDER_GAIN; the shipped checkpoint's layer 0).  Eval mode only.
  * `model.node_order` ("auto" | "input" | "locality", default "auto"): graph_parser.py:174-181 numbers reads in S-line order, which need not be
    the layout's; a cached graph object whose mean edge span says so gets its nodes renumbered once (gnnome_amd/node_order.py) - callers
    never see it (x goes in and logits come out in their numbering).
  * `model.activation_storage` ("fp32" | "bf16", default "fp32"; train mode only): "bf16" keeps the pre-normalisation gate output
    and its gradient in HBM as bfloat16 between the kernels of the training step (gnnome_amd/train.py; arithmetic stays fp32).
"""
import torch.nn as nn

from . import engine
from .layers import ScorePredictor, SymGatedGCN_processor


class SymGatedGCNModel(nn.Module):
    arithmetic = "auto"
    activation_storage = "fp32"
    node_order = "auto"      # gnnome_amd.graph.views_for: renumber the nodes of a cached graph object whose ids do not follow the layout
    range_check = True       # engine.forward_in_range: a forward that left fp16x3's operand range is run again as bf16x6

    def __init__(self, node_features, edge_features, hidden_features, hidden_ne_features, num_layers,
                 hidden_edge_scores, normalization, dropout=None):
        super().__init__()
        if not (1 <= node_features <= 8 and 1 <= edge_features <= 8 and 1 <= hidden_ne_features <= 64):
            raise ValueError("the encoder kernels take node/edge features <= 8 and hidden_ne_features <= 64 "
                             "(reference: 2, 2, 16 - configs/hyperparameters.py:20-25)")
        self.linear1_node = nn.Linear(node_features, hidden_ne_features, bias=True)
        self.linear2_node = nn.Linear(hidden_ne_features, hidden_features, bias=True)
        self.linear1_edge = nn.Linear(edge_features, hidden_ne_features, bias=True)
        self.linear2_edge = nn.Linear(hidden_ne_features, hidden_features, bias=True)
        self.gnn = SymGatedGCN_processor(num_layers, hidden_features, normalization, dropout=dropout)
        self.predictor = ScorePredictor(hidden_features, hidden_edge_scores)
        self.relu = nn.ReLU()

    def forward(self, graph, x, e):
        return engine.model_forward(self, graph, x, e)


__all__ = ["SymGatedGCNModel"]
