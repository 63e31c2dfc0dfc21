"""The fp16x3 arithmetic of gnnome_amd/csrc/edge_tile_f16.hip, restated in numpy and measured against an fp64 product (CPU only).

x = x1 + x2 / 2048 + rx with x1 = RN16(x), x2 = RN16((x - x1) * 2048); products x1 w1 (main accumulator) and x1 w2 + x2 w1 (second
accumulator, folded in with 2^-11 once per tile); fp32 accumulation in 16-wide k steps as the matrix cores do.  The claims of the kernel's
header are checked here: the split is exact to 2^-22, the product is no further from the exact one than an fp32 product (the reference's
arithmetic: torch CPU / MKL, layers/gated_gcn_full.py:91-97) and than the bf16x6 form it replaces, scaling by powers of two commutes with
it, and operands far below 1 keep their precision because the second plane is stored scaled."""
import numpy as np
import torch

K, M, N = 256, 1024, 256


def split_f16(a):
    a1 = a.astype(np.float16)
    r = ((a - a1.astype(np.float32)) * np.float32(2048)).astype(np.float32)      # exact: a - a1 fits fp32, so does its 2^11-fold
    return a1.astype(np.float32), r.astype(np.float16).astype(np.float32)


def trunc_bf16(a):
    return (a.view(np.uint32) & np.uint32(0xFFFF0000)).view(np.float32)


def split_bf16(a):
    h = trunc_bf16(a)
    r = (a - h).astype(np.float32)
    m = trunc_bf16(r)
    return h, m, (r - m).astype(np.float32)


def chain(terms, k=K):
    """fp32 accumulator over 16-wide k steps; within a step the products are exact and summed before one rounding (a model of one MFMA)."""
    out = np.zeros((terms[0][0].shape[0], terms[0][1].shape[0]), np.float32)
    for k0 in range(0, k, 16):
        for a, b in terms:
            out = (out.astype(np.float64) + a[:, k0:k0 + 16].astype(np.float64) @ b[:, k0:k0 + 16].astype(np.float64).T).astype(np.float32)
    return out


def f16x3(x, w):
    x1, x2 = split_f16(x)
    w1, w2 = split_f16(w)
    main, corr = chain([(x1, w1)]), chain([(x1, w2), (x2, w1)])
    return (main + corr * np.float32(1 / 2048)).astype(np.float32)


def bf16x6(x, w):
    xh, xm, xl = split_bf16(x)
    wh, wm, wl = split_bf16(w)
    return chain([(xl, wh), (xh, wl), (xm, wm), (xm, wh), (xh, wm), (xh, wh)])


def _operands(seed, xs=3.0):
    rng = np.random.default_rng(seed)
    x = (rng.standard_normal((M, K)) * xs).astype(np.float32)
    w = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    return x, w


def _err(y, x, w):
    ref = x.astype(np.float64) @ w.astype(np.float64).T
    scale = np.abs(x).astype(np.float64) @ np.abs(w).astype(np.float64).T
    return float(np.max(np.abs(y.astype(np.float64) - ref) / scale))


def test_two_fp16_planes_hold_an_fp32_value_to_2_pow_minus_22():
    x, _ = _operands(0)
    x1, x2 = split_f16(x)
    back = x1.astype(np.float64) + x2.astype(np.float64) / 2048
    assert np.all(np.abs(back - x) <= 2.0 ** -22 * np.abs(x) + 2.0 ** -36)      # (the absolute term: |x| below fp16's normal range, 6e-5)
    assert np.all(np.abs(x2) <= 1.0001 * np.abs(x1) + 1e-30)      # the scaled second plane is never larger than the first: same fp16 range


def test_f16x3_is_no_further_from_the_exact_product_than_fp32_and_bf16x6():
    x, w = _operands(1)
    e16, e6 = _err(f16x3(x, w), x, w), _err(bf16x6(x, w), x, w)
    e32 = _err((torch.from_numpy(x) @ torch.from_numpy(w).T).numpy(), x, w)
    assert e16 <= 3 * 2.0 ** -22 and e16 <= e32 and e16 <= e6, (e16, e6, e32)


def test_powers_of_two_commute_and_small_operands_keep_their_precision():
    x, w = _operands(2)
    x = (np.sign(x) * (np.abs(x) + np.float32(2.0 ** -6))).astype(np.float32)      # every operand inside fp16's normal range at both scales
    w = (np.sign(w) * (np.abs(w) + np.float32(2.0 ** -8))).astype(np.float32)
    y = f16x3(x, w)
    assert np.array_equal(f16x3(x * np.float32(64), w), y * np.float32(64))       # no fp16 over- or underflow in between: bit for bit
    small = (x * np.float32(2.0 ** -12)).astype(np.float32)                       # |x| ~ 7e-4: fp16 would be denormal for the UNSCALED second plane
    assert _err(f16x3(small, w), small, w) <= 3 * 2.0 ** -22
    tiny = (x * np.float32(2.0 ** -24)).astype(np.float32)                        # |x| ~ 2e-7, far below fp16's normal range: absolute error 2^-36 per operand
    ref = tiny.astype(np.float64) @ w.astype(np.float64).T
    assert np.max(np.abs(f16x3(tiny, w) - ref)) <= K * 2.0 ** -36
    big = np.full((4, K), 70000.0, np.float32)
    assert not np.isfinite(f16x3(big, w)).any()                                   # beyond fp16's range the result is loud, not wrong


def test_whole_model_probabilities_with_emulated_products():
    """DESIGN.md section 4's accuracy table, extended by fp16x3: the oracle's 8-layer model (models/full_graph.py:22-30 restated) with the six
    H x H products of every layer replaced by the emulated split arithmetic, against an fp64 evaluation of the same model - max |dp| on
    the edge probabilities is what BASELINE's 1e-4 bar is about.  fp16x3 must be no worse than the fp32 model and than bf16x6."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from gnnome_amd.synth import make_graph, random_state_dict
    from oracle.symgated_oracle import degree_features, model_from_state_dict
    n, e, hidden = 2000, 20_000, 128
    g = make_graph(n, e, seed=3)
    x = degree_features(g["src"], g["dst"], n)
    sd = random_state_dict(hidden, seed=3)
    graph = (g["src"], g["dst"], n)
    with torch.no_grad():
        p64 = torch.sigmoid(model_from_state_dict(sd, dtype=torch.float64).eval()(graph, x.double(), g["e"].double())).squeeze(-1)

    def run(product):
        m = model_from_state_dict(sd).eval()
        if product is not None:
            for conv in m.gnn.convs:
                for name in ("A_1", "A_2", "A_3", "B_1", "B_2", "B_3"):
                    lin = getattr(conv, name)
                    w, b = lin.weight.detach().numpy(), lin.bias.detach()
                    lin.forward = (lambda inp, w=w, b=b: torch.from_numpy(product(inp.numpy(), w)) + b)
        with torch.no_grad():
            p = torch.sigmoid(m(graph, x, g["e"])).squeeze(-1)
        return (p.double() - p64).abs().max().item()

    def k128(fn):
        return lambda a, w: fn(np.ascontiguousarray(a, dtype=np.float32), np.ascontiguousarray(w, dtype=np.float32))

    global K
    old_k, K = K, hidden
    try:
        d32, d6, d16 = run(None), run(k128(bf16x6)), run(k128(f16x3))
    finally:
        K = old_k
    print(f"max |dp| against fp64: fp32 model {d32:.2e}, bf16x6 {d6:.2e}, fp16x3 {d16:.2e}")
    assert d16 <= 2 * max(d32, d6) + 1e-7 and d16 < 1e-5
