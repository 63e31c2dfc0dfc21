"""Generate tests/golden/g8_layernorm_train_h64.pt from the REFERENCE's own classes.  Build container only:

    python tests/golden/make_golden_ln_train.py       # needs /root/reference (read-only)

normalization='layer' (layers/gated_gcn_full.py:40-42) in TRAIN mode, one BCE step under autograd: loss, logits and
the gradient of every parameter.  Same harness as make_golden.py (reference models.SymGatedGCNModel, the DGL test
double tests/golden/_dgl_shim); the fixture is data only."""
import os

import torch
import torch.nn.functional as F

import make_golden as mg   # sets sys.path for the reference + shim, provides ref_model / run / save
from gnnome_amd.synth import make_graph, random_state_dict
from oracle.symgated_oracle import degree_features


def main():
    torch.set_num_threads(1)   # as make_golden.py: the reduction order of torch's CPU kernels depends on the thread count
    n, ec = 150, 1500
    gr = make_graph(n, ec, seed=8, kind="banded")
    src, dst, e, y, pw = gr["src"], gr["dst"], gr["e"], gr["y"], gr["pos_weight"]
    x = degree_features(src, dst, n)
    sd = {k: v for k, v in random_state_dict(64, seed=4).items() if "running_" not in k and "num_batches" not in k}
    m = mg.ref_model(sd, 64, normalization="layer", dropout=0.0).train()
    out, _ = mg.run(m, src, dst, n, x, e)
    loss = F.binary_cross_entropy_with_logits(out.squeeze(-1), y, pos_weight=pw)
    loss.backward()
    grads = {k: p.grad.clone() for k, p in m.named_parameters()}
    mg.save("g8_layernorm_train_h64.pt", dict(src=src, dst=dst, num_nodes=n, x=x, e=e, y=y, pos_weight=pw, seed=4, hidden=64,
                                              logits=out.detach(), loss=loss.detach(), grads=grads))


if __name__ == "__main__":
    main()
