"""Generate tests/golden/g12_layernorm_widths.pt from the REFERENCE's own classes.  Build container only:

    python tests/golden/make_golden_ln_widths.py       # needs /root/reference (read-only)

normalization='layer' (layers/gated_gcn_full.py:40-42) at widths BETWEEN the ones the HIP kernels are built for - (hidden_features,
hidden_edge_scores) = (96, 48) and (160, 128): eval-mode logits, and one train-mode BCE step under autograd (loss, logits, the gradient
of every parameter).  LayerNorm's statistics run over the row, so the zero-padding that serves BatchNorm models needs kernels that
normalise over the model's own width (GNNOME_NORM_LAYER_OVER, round 5).  hidden_edge_scores = 128 also pins the scorer's backward at
its widest.  Same harness as make_golden.py (reference models.SymGatedGCNModel, the DGL test double tests/golden/_dgl_shim); the
fixture is data only."""
import torch
import torch.nn.functional as F

import make_golden as mg   # sets sys.path for the reference + shim, provides ref_model / run / save
from gnnome_amd.synth import make_graph, random_state_dict
from oracle.symgated_oracle import degree_features

CASES = ((96, 48), (160, 128))


def main():
    torch.set_num_threads(1)   # as make_golden.py: the reduction order of torch's CPU kernels depends on the thread count
    n, ec, layers = 150, 1500, 2
    gr = make_graph(n, ec, seed=8, kind="banded")
    src, dst, e, y, pw = gr["src"], gr["dst"], gr["e"], gr["y"], gr["pos_weight"]
    x = degree_features(src, dst, n)
    cases = []
    for hidden, hs in CASES:
        sd = {k: v for k, v in random_state_dict(hidden, num_layers=layers, hidden_edge_scores=hs, seed=6).items()
              if "running_" not in k and "num_batches" not in k}
        m = mg.ref_model(sd, hidden, normalization="layer", dropout=0.0, layers=layers, hs=hs).eval()
        with torch.no_grad():
            eval_logits, _ = mg.run(m, src, dst, n, x, e)
        m.train()
        out, _ = mg.run(m, src, dst, n, x, e)
        loss = F.binary_cross_entropy_with_logits(out.squeeze(-1), y, pos_weight=pw)
        loss.backward()
        cases.append(dict(hidden=hidden, hs=hs, layers=layers, seed=6, eval_logits=eval_logits.detach(), logits=out.detach(), loss=loss.detach(),
                          grads={k: p.grad.clone() for k, p in m.named_parameters()}))
    mg.save("g12_layernorm_widths.pt", dict(src=src, dst=dst, num_nodes=n, x=x, e=e, y=y, pos_weight=pw, cases=cases))


if __name__ == "__main__":
    main()
