"""Generate tests/golden/g7_closure.pt from the REFERENCE's own caller-side functions.  Build container only:

    python tests/golden/make_golden_closure.py        # needs /root/reference (read-only)

The functions either side of the model call that SURVEY.md 8f ranks next:
  * train.py:103-109     symmetry_loss                       (value + autograd gradients w.r.t. both logit vectors)
  * train.py:138-145     the BCE-with-logits(pos_weight) of get_bce_loss_full  (F.binary_cross_entropy_with_logits)
  * train.py:112-122     get_full_ne_features                (z-scored degrees, both orientations)
  * utils/data_utils.py:31-41  preprocess_graph              (z-scored overlap length | similarity)
  * utils/metrics.py:6-12, 15-28  calculate_tfpn, calculate_metrics
`train.py` and `utils/data_utils.py` import packages this image lacks (wandb, DGL datasets, Bio), so their
functions are compiled one at a time from the reference file's syntax tree and executed here with torch in scope -
the reference's text runs, none of it is stored.  utils/metrics.py imports as is.  The fixture is data only.
"""
import ast
import os
import sys

import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REF)
sys.path.insert(1, ROOT)

from utils import metrics as ref_metrics  # noqa: E402  (the reference's utils/metrics.py)
from configs.hyperparameters import get_hyperparameters  # noqa: E402  (the reference's configs/hyperparameters.py)

from gnnome_amd.synth import make_graph  # noqa: E402


def reference_functions(path, names, scope):
    tree = ast.parse(open(path).read())
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in names:
            exec(compile(ast.Module([node], []), path, "exec"), scope)
    return [scope[n] for n in names]


class FakeGraph:
    """What the two feature functions touch of a DGLGraph: ndata / edata dicts, int(), num_nodes()."""

    def __init__(self, n, ndata, edata):
        self.n, self.ndata, self.edata = n, ndata, edata

    def int(self):
        return self

    def num_nodes(self):
        return self.n


def main():
    symmetry_loss, get_full_ne_features = reference_functions(os.path.join(REF, "train.py"), ["symmetry_loss", "get_full_ne_features"],
                                                              {"torch": torch, "F": F})
    (preprocess_graph,) = reference_functions(os.path.join(REF, "utils", "data_utils.py"), ["preprocess_graph"],
                                              {"torch": torch, "get_hyperparameters": get_hyperparameters})
    gen = torch.Generator().manual_seed(7)
    E = 20_000
    org = (3.0 * torch.randn(E, generator=gen)).requires_grad_()
    rev = (org.detach() + 0.5 * torch.randn(E, generator=gen)).requires_grad_()
    with torch.no_grad():   # exact ties (|a-b| has sign 0 there), large logits of both signs, an exact zero
        rev[:100] = org[:100]
        org[100:110] = torch.tensor([40., -40., 90., -90., 0., 1e-9, -1e-9, 15., -15., 0.])
        rev[100:110] = torch.tensor([-40., 40., 90., -90., 0., 0., 0., -15., 15., 5.])
    labels = (torch.rand(E, generator=gen) < 0.3).float()
    pos_weight = torch.tensor(0.7 / 0.3)
    alpha = 0.1
    sym = symmetry_loss(org, rev, labels, pos_weight, alpha=alpha)
    sym.backward()
    sym_grads = (org.grad.clone(), rev.grad.clone())
    org.grad = None
    bce = F.binary_cross_entropy_with_logits(org, labels, pos_weight=pos_weight)   # train.py:144
    bce.backward()
    tfpn = ref_metrics.calculate_tfpn(org.detach(), labels)
    tfpn_rev = ref_metrics.calculate_tfpn(rev.detach(), labels)

    n, e_cnt = 2000, 20_000
    gr = make_graph(n, e_cnt, seed=11, kind="banded")
    src, dst = gr["src"].long(), gr["dst"].long()
    ndata = {"in_deg": torch.bincount(dst, minlength=n).float(), "out_deg": torch.bincount(src, minlength=n).float()}
    ol_len = torch.randint(500, 30_000, (e_cnt,), generator=gen)       # overlap lengths in bases (int64, as the parser stores them)
    ol_sim = 0.9 + 0.1 * torch.rand(e_cnt, generator=gen)
    g = preprocess_graph(FakeGraph(n, ndata, {"overlap_length": ol_len, "overlap_similarity": ol_sim}))
    x_fwd, e_feat = get_full_ne_features(g, reverse=False)
    x_rev, _ = get_full_ne_features(g, reverse=True)

    out = dict(
        org=org.detach(), rev=rev.detach(), labels=labels, pos_weight=pos_weight, alpha=alpha,
        symmetry_loss=sym.detach(), symmetry_grad_org=sym_grads[0], symmetry_grad_rev=sym_grads[1],
        bce_loss=bce.detach(), bce_grad=org.grad.clone(),
        tfpn=tfpn, tfpn_rev=tfpn_rev, metrics=ref_metrics.calculate_metrics(*tfpn),
        metrics_inverse=ref_metrics.calculate_metrics_inverse(*tfpn),
        src=gr["src"], dst=gr["dst"], num_nodes=n, overlap_length=ol_len, overlap_similarity=ol_sim,
        x=x_fwd, x_reversed=x_rev, e=e_feat,
    )
    path = os.path.join(HERE, "g7_closure.pt")
    torch.save(out, path)
    print(f"wrote {path} ({os.path.getsize(path) / 1e6:.2f} MB)  sym={sym.item():.6f} bce={bce.item():.6f} tfpn={tfpn}")


if __name__ == "__main__":
    main()
