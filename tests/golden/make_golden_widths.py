"""Generate tests/golden/g11_widths.pt from the REFERENCE's own classes (build container only; see make_golden.py):

    python tests/golden/make_golden_widths.py

Widths BETWEEN the ones the HIP kernels are built for - the reference takes any hidden_features / hidden_edge_scores
(configs/hyperparameters.py:22-24) - on one small banded graph, eval mode, random init with non-trivial BatchNorm buffers
(`gnnome_amd.synth.random_state_dict`, regenerated from the seed by the tests).  The fixture is data: inputs + the reference's logits.
"""
import torch

from make_golden import ref_model, run, save  # (sets sys.path for the DGL shim and the reference)

from gnnome_amd.synth import make_graph, random_state_dict
from oracle.symgated_oracle import degree_features

CASES = ((40, 20), (96, 48), (200, 100), (128, 40), (72, 64))   # (hidden_features, hidden_edge_scores)


def main():
    torch.set_num_threads(1)
    torch.manual_seed(0)
    n, ec = 300, 3000
    gr = make_graph(n, ec, seed=11, kind="banded")
    src, dst, e = gr["src"], gr["dst"], gr["e"]
    x = degree_features(src, dst, n)
    cases = []
    for hidden, hs in CASES:
        sd = random_state_dict(hidden, num_layers=4, hidden_edge_scores=hs, seed=11)
        m = ref_model(sd, hidden, layers=4, hs=hs).eval()
        with torch.no_grad():
            out, tr = run(m, src, dst, n, x, e, trace=True)
        cases.append(dict(hidden=hidden, hs=hs, layers=4, seed=11, logits=out, h_final_rows=tr[-1][0][:32].clone(), e_final_rows=tr[-1][1][:32].clone()))
    save("g11_widths.pt", dict(src=src, dst=dst, num_nodes=n, x=x, e=e, cases=cases))


if __name__ == "__main__":
    main()
