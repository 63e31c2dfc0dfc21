"""Generate tests/golden/*.pt from the REFERENCE's own classes.  Run in the build container only:

    python tests/golden/make_golden.py            # needs /root/reference (read-only)

It imports the reference's unmodified `models.SymGatedGCNModel` (models/full_graph.py:9-30 ->
layers/processor.py:9-19 -> layers/gated_gcn_full.py:8-142 -> layers/score_predictor.py:5-24) with
`tests/golden/_dgl_shim` standing in for the un-installable DGL 0.8.1 wheel, runs it on small seeded
inputs and stores inputs + outputs.  The fixtures are data; no reference source is copied.
Weights: `weights.pt` is the reference's shipped state_dict (BSD-3 data file, weights/weights.pt);
random-init cases regenerate theirs from `gnnome_amd.synth.random_state_dict(seed)` instead of
storing them.
"""
import io
import os
import shutil
import sys
from contextlib import redirect_stdout

import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, os.path.join(HERE, "_dgl_shim"))
sys.path.insert(1, REF)
sys.path.insert(2, ROOT)

import dgl  # noqa: E402  (the shim)
import models  # noqa: E402  (the reference)

from gnnome_amd.synth import make_graph, random_state_dict  # noqa: E402
from oracle.symgated_oracle import degree_features  # noqa: E402


def ref_model(sd, hidden, normalization="batch", dropout=None, layers=8, hs=64):
    m = models.SymGatedGCNModel(2, 2, hidden, 16, layers, hs, normalization, dropout=dropout)
    m.load_state_dict(sd)
    return m


def run(m, src, dst, n, x, e, trace=False):
    g = dgl.graph((src.long(), dst.long()), num_nodes=n)
    tr = []
    hooks = []
    if trace:
        for conv in m.gnn.convs:
            hooks.append(conv.register_forward_hook(lambda mod, i, o: tr.append((o[0].detach().clone(), o[1].detach().clone()))))
    with redirect_stdout(io.StringIO()):  # models/full_graph.py:25 prints x.shape
        out = m(g, x, e)
    for h in hooks:
        h.remove()
    assert not g.ndata and not g.edata, "local_scope must leave the graph untouched"
    return out, tr


def uniform_graph(n, e_cnt, seed):
    g = torch.Generator().manual_seed(seed)
    src = torch.randint(0, n, (e_cnt,), generator=g, dtype=torch.int64).int()
    dst = torch.randint(0, n, (e_cnt,), generator=g, dtype=torch.int64).int()
    ef = torch.stack([torch.randn(e_cnt, generator=g), 0.9 + 0.1 * torch.rand(e_cnt, generator=g)], 1)
    return src, dst, ef


def save(name, obj):
    path = os.path.join(HERE, name)
    torch.save(obj, path)
    print(f"{name}: {os.path.getsize(path) / 1024:.0f} KiB")


def main():
    torch.set_num_threads(1)
    torch.manual_seed(0)
    shutil.copyfile(os.path.join(REF, "weights", "weights.pt"), os.path.join(HERE, "weights.pt"))
    wsd = torch.load(os.path.join(HERE, "weights.pt"), map_location="cpu")

    # G1: hand graph - node 6 has no in-edges, node 7 no out-edges, node 5 isolated, a self-loop, a duplicate edge
    src = torch.tensor([0, 1, 2, 3, 4, 0, 1, 2, 2, 3, 6, 6, 4, 1], dtype=torch.int32)
    dst = torch.tensor([1, 2, 3, 4, 0, 2, 3, 2, 7, 7, 0, 1, 7, 2], dtype=torch.int32)
    n = 8
    x = degree_features(src, dst, n)
    g = torch.Generator().manual_seed(11)
    e = torch.stack([torch.randn(14, generator=g), 0.9 + 0.1 * torch.rand(14, generator=g)], 1)
    m = ref_model(wsd, 64).eval()
    with torch.no_grad():
        out, tr = run(m, src, dst, n, x, e, trace=True)
    save("g1_hand.pt", dict(src=src, dst=dst, num_nodes=n, x=x, e=e, logits=out,
                            h_final=tr[-1][0], e_final=tr[-1][1], weights="weights.pt"))

    # G2: N=1000, E=10000 uniform, shipped weights, eval, per-layer digests
    n, ec = 1000, 10000
    src, dst, e = uniform_graph(n, ec, 0)
    x = degree_features(src, dst, n)
    with torch.no_grad():
        out, tr = run(m, src, dst, n, x, e, trace=True)
    layers = [dict(h_sum=h.double().sum().item(), e_sum=ee.double().sum().item(), h_rows=h[:8].clone(),
                   e_rows=ee[:8].clone()) for h, ee in tr]
    save("g2_uniform_1k.pt", dict(src=src, dst=dst, num_nodes=n, x=x, e=e, logits=out, layers=layers,
                                  weights="weights.pt"))

    # G3 + G4: train mode, random init H=64, dropout=0, BCE(pos_weight); reversed pass + symmetry loss
    n, ec = 200, 2000
    gr = make_graph(n, ec, seed=3, kind="banded")
    src, dst, e, y, pw = gr["src"], gr["dst"], gr["e"], gr["y"], gr["pos_weight"]
    x = degree_features(src, dst, n)
    sd = random_state_dict(64, seed=1)
    m3 = ref_model(sd, 64, dropout=0.0).train()
    out, _ = run(m3, src, dst, n, x, e)
    loss = F.binary_cross_entropy_with_logits(out.squeeze(-1), y, pos_weight=pw)
    loss.backward()
    grads = {k: p.grad.clone() for k, p in m3.named_parameters()}
    buffers = {k: b.clone() for k, b in m3.named_buffers()}
    save("g3_train_h64.pt", dict(src=src, dst=dst, num_nodes=n, x=x, e=e, y=y, pos_weight=pw, seed=1, hidden=64,
                                 logits=out.detach(), loss=loss.detach(), grads=grads, buffers_after=buffers))
    m4 = ref_model(sd, 64, dropout=0.0).eval()
    with torch.no_grad():
        org, _ = run(m4, src, dst, n, x, e)
        x_rev = degree_features(src, dst, n, reverse=False)[:, [1, 0]]  # train.py:116-117 swaps the columns
        rev, _ = run(m4, dst, src, n, x_rev, e)  # dgl.reverse(g, True, True): endpoints swapped, ids kept
        o, r = org.squeeze(-1), rev.squeeze(-1)
        sym = (F.binary_cross_entropy_with_logits(o, y, pos_weight=pw, reduction="none")
               + F.binary_cross_entropy_with_logits(r, y, pos_weight=pw, reduction="none")
               + 0.1 * torch.abs(o - r)).mean()
    save("g4_reverse_h64.pt", dict(src=src, dst=dst, num_nodes=n, x=x, x_rev=x_rev, e=e, y=y, pos_weight=pw, seed=1,
                                   hidden=64, alpha=0.1, logits=org, logits_rev=rev, symmetry_loss=sym))

    # G5: H=128 / H=256 random init, eval
    n, ec = 500, 5000
    gr = make_graph(n, ec, seed=5, kind="banded")
    src, dst, e = gr["src"], gr["dst"], gr["e"]
    x = degree_features(src, dst, n)
    for hidden in (128, 256):
        m5 = ref_model(random_state_dict(hidden, seed=1), hidden).eval()
        with torch.no_grad():
            out, tr = run(m5, src, dst, n, x, e, trace=True)
        save(f"g5_eval_h{hidden}.pt", dict(src=src, dst=dst, num_nodes=n, x=x, e=e, seed=1, hidden=hidden, logits=out,
                                           h_final=tr[-1][0], e_final_rows=tr[-1][1][:64].clone()))

    # G6: LayerNorm variant, H=64
    m6 = ref_model({k: v for k, v in random_state_dict(64, seed=2).items()
                    if "running_" not in k and "num_batches" not in k}, 64, normalization="layer").eval()
    with torch.no_grad():
        out, _ = run(m6, src, dst, n, x, e)
    save("g6_layernorm_h64.pt", dict(src=src, dst=dst, num_nodes=n, x=x, e=e, seed=2, hidden=64, logits=out))


if __name__ == "__main__":
    main()
