"""Reference-ORDER arithmetic (csrc/reference_order.hip, engine.REFERENCE_ORDER_GAIN) and the parity configuration that
needs it: an E. coli-sized graph (N = 30 000, E = 300 000, H = 64) with the SHIPPED weights (the substitute
for BASELINE configs[0]).

What is pinned here
  * CPU: the two facts the kernels are built on - torch's CPU nn.Linear is one k-ascending fma chain from zero with the
    bias added afterwards, and its eval-mode BatchNorm1d is fma(x, alpha, beta) - are checked on the machine that runs
    the tests (tools/cpu_linear_order_probe.py prints the table: bit-identical for every row on the build container's
    Intel AVX-512 host from M = 5 up; on the GPU pool's AMD EPYC hosts for every row inside MKL's full row panels -
    the ragged last rows of a matrix go through another MKL kernel there); the host logic (which layers go through the
    reference-order entries) is replayed on the checker backend against the fp32 oracle.
  * GPU: the reference-order kernels equal the k-ordered chain BIT FOR BIT (the chain is emulated through fp64, so the
    expected value does not depend on the host's BLAS); the whole model on the E. coli-sized graph is compared with
    (i) the fp32 torch oracle - north_star's bar, < 1e-4 on edge probabilities - and (ii) the fp64 C oracle; the
    measured distances are appended to gpurun_out/parity_margins.jsonl.
"""
import json
import os

import pytest
import torch
import torch.nn.functional as F

import cpu_ops
import gnnome_amd
from conftest import ROOT, load_golden
from gnnome_amd import engine
from gnnome_amd.synth import make_graph
from oracle.symgated_oracle import degree_features, model_from_state_dict

PROB_TOL = 1e-4          # north_star: max |sigmoid(logit) - sigmoid(reference logit)|
AUTO_TOL = 2e-5          # what the auto arithmetic is expected to hold against the fp32 oracle (measured: ~2e-6)
ECOLI = (30_000, 300_000)


def _dprob(a, b):
    return (torch.sigmoid(a.double().cpu()) - torch.sigmoid(b.double().cpu())).abs().max().item()


def _chain(A, W, b=None):
    """k-ascending fma chain from zero, bias afterwards, emulated through fp64 (exact products, one rounding per step;
    the double rounding differs from a true fma once in ~2^29 elements)."""
    acc = torch.zeros(A.shape[0], W.shape[0], dtype=torch.float64)
    for k in range(A.shape[1]):
        acc = (acc + A[:, k:k + 1].double() * W[:, k].double().unsqueeze(0)).float().double()
    return (acc if b is None else acc + b.double()).float()


def _same_bits(got, want):
    got = got.cpu()
    return got.shape == want.shape and (got != want).float().mean().item() < 2e-6


def _record(name, **values):
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "parity_margins.jsonl"), "a") as f:
        f.write(json.dumps({"test": name, **values}) + "\n")


# ------------------------------------------------------------------------------------------------ CPU: the two facts

@pytest.mark.parametrize("threads", [1, 8])
def test_torch_cpu_linear_is_a_k_ordered_fma_chain(threads):
    old = torch.get_num_threads()
    torch.set_num_threads(threads)
    try:
        g = torch.Generator().manual_seed(0)
        for M, K, N in [(768, 2, 16), (768, 16, 64), (5120, 64, 64), (1024, 64, 320), (1024, 128, 640), (512, 192, 64)]:
            A, W, b = 10 * torch.randn(M, K, generator=g), 0.3 * torch.randn(N, K, generator=g), torch.randn(N, generator=g)
            # a blocked, bias-first or unfused evaluation differs in 55-75 % of the elements; MKL's ragged-row kernels
            # (a few rows per matrix on some hosts) are the only tolerated exception
            assert (F.linear(A, W, b) != _chain(A, W, b)).float().mean().item() < 0.02, (M, K, N)
    finally:
        torch.set_num_threads(old)


def test_torch_eval_batchnorm_is_one_fma_and_engine_folds_it_the_same_way():
    g = torch.Generator().manual_seed(1)
    H = 64
    bn = torch.nn.BatchNorm1d(H).eval()
    with torch.no_grad():
        bn.running_var.copy_(torch.rand(H, generator=g) * 50 + 1e-4)
        bn.running_mean.copy_(torch.randn(H, generator=g) * 5)
        bn.weight.copy_(torch.randn(H, generator=g))
        bn.bias.copy_(torch.randn(H, generator=g))
    x = 100 * torch.randn(50_000, H, generator=g)
    kind, scale, shift = engine._norm_affine(bn, torch.device("cpu"))
    assert kind == engine.NORM_AFFINE
    with torch.no_grad():
        want = bn(x)
    got = cpu_ops._fma(x, scale, shift)
    assert (got != want).float().mean().item() < 1e-6   # (the fp64 emulation of fma double-rounds once in ~2^29)


# ------------------------------------------------------------------------------------------------ CPU: host logic

def _prepared(sd, arithmetic, device):
    m = gnnome_amd.SymGatedGCNModel(2, 2, 64, 16, 8, 64, "batch").eval()
    m.load_state_dict(sd)
    m.arithmetic = arithmetic
    return m, engine.Prepared(m, device)


def test_auto_arithmetic_picks_the_high_gain_layer(shipped_weights):
    _, prep = _prepared(shipped_weights, "auto", torch.device("cpu"))
    gains = [lw.gain_e for lw in prep.layers]
    assert gains[0] > 100 and max(gains[1:]) < engine.REFERENCE_ORDER_GAIN, gains
    assert [lw.ref for lw in prep.layers] == [True] + [False] * 7
    assert all(lw.ref for lw in _prepared(shipped_weights, "reference", torch.device("cpu"))[1].layers)
    assert not any(lw.ref for lw in _prepared(shipped_weights, "fast", torch.device("cpu"))[1].layers)
    with pytest.raises(ValueError):
        _prepared(shipped_weights, "exact", torch.device("cpu"))


def test_checker_backend_ecoli_size_against_the_fp32_oracle(shipped_weights):
    """The kernel sequence on the checker backend: with layer 0 in the reference's order the model lands ~2e-6 from the
    fp32 oracle; with every layer reordered ('fast') it lands at the model's fp32 noise floor, above 1e-4 on this graph."""
    n, e = ECOLI
    gr = make_graph(n, e, seed=1, kind="banded")
    x = degree_features(gr["src"], gr["dst"], n)
    with torch.no_grad():
        want = model_from_state_dict(shipped_weights).eval()((gr["src"], gr["dst"], n), x, gr["e"])
    views = cpu_ops.CpuViews(gr["src"], gr["dst"], n)
    dist = {}
    for mode in ("auto", "fast"):
        _, prep = _prepared(shipped_weights, mode, torch.device("cpu"))
        with torch.no_grad():
            dist[mode] = _dprob(engine.run_stack(cpu_ops, prep, views, x, gr["e"]).unsqueeze(1), want)
    assert dist["auto"] < AUTO_TOL, dist
    assert dist["fast"] > 5 * dist["auto"], dist   # the reason the reference-order kernels exist


# ------------------------------------------------------------------------------------------------ GPU

def dev():
    return torch.device("cuda", 0)


@pytest.mark.gpu
@pytest.mark.parametrize("m,k,nout", [(1000, 64, 320), (30_001, 64, 320), (777, 128, 640), (5, 64, 8), (4097, 128, 128), (2049, 256, 1280), (33, 256, 128)])
def test_linear_ref_equals_torch_cpu_bit_for_bit(m, k, nout):
    from gnnome_amd import ops
    g = torch.Generator().manual_seed(m + k)
    A, W, b = 10 * torch.randn(m, k, generator=g), 0.3 * torch.randn(nout, k, generator=g), torch.randn(nout, generator=g)
    want = _chain(A, W, b)
    assert _same_bits(ops.linear_ref(A.to(dev()), W.to(dev()), b.to(dev())), want)
    assert _same_bits(ops.linear_ref(A.to(dev()), W.to(dev()), None), _chain(A, W))
    if m >= 64 and k <= 128:   # and that IS what torch's CPU nn.Linear returns (MKL's full row panels; see the module docstring)
        assert (F.linear(A, W, b) != want).float().mean().item() < 0.02
    # (K = 256, round 4: whether MKL keeps the 256 products of an output in ONE chain depends on the host - the build container's Intel
    #  host does, bit for bit; the GPU pool's EPYC hosts split k into panels and 88 % of the outputs differ from the chain.  At this width
    #  "the reference's fp32 result" is host-dependent; the kernels implement the single chain, which is also what K <= 128 gives everywhere.)
    # strided output block and strided weight rows (the [N,5H] projection, column blocks of predictor.W1)
    P = torch.zeros(m, nout + 64, device=dev())
    ops.linear_ref(A.to(dev()), W.to(dev()), b.to(dev()), out=P[:, 64:])
    assert _same_bits(P[:, 64:], want) and not P[:, :64].any()
    W2 = torch.randn(nout, 3 * k, generator=g)
    assert _same_bits(ops.linear_ref(A.to(dev()), W2.to(dev())[:, k:2 * k], None), _chain(A, W2[:, k:2 * k]))


@pytest.mark.gpu
@pytest.mark.parametrize("hidden", [64, 128])
def test_encoder_equals_torch_cpu_bit_for_bit(hidden):
    from gnnome_amd import ops
    g = torch.Generator().manual_seed(hidden)
    x = torch.randn(3001, 2, generator=g)
    W1, b1 = torch.randn(16, 2, generator=g), torch.randn(16, generator=g)
    W2, b2 = torch.randn(hidden, 16, generator=g), torch.randn(hidden, generator=g)
    perm = torch.randperm(3001, generator=g).int()
    want = _chain(torch.relu(_chain(x, W1, b1)), W2, b2)
    assert _same_bits(ops.encode(*(t.to(dev()) for t in (x, W1, b1, W2, b2))), want)
    assert _same_bits(ops.encode(*(t.to(dev()) for t in (x, W1, b1, W2, b2)), gather=perm.to(dev())), want[perm.long()])
    assert (F.linear(torch.relu(F.linear(x, W1, b1)), W2, b2) != want).float().mean().item() < 0.02   # = torch on the CPU
    # the general kernel (other in_features / hidden_ne)
    x5 = torch.randn(500, 5, generator=g)
    W1b, b1b = torch.randn(40, 5, generator=g), torch.randn(40, generator=g)
    W2b, b2b = torch.randn(hidden, 40, generator=g), torch.randn(hidden, generator=g)
    want = _chain(torch.relu(_chain(x5, W1b, b1b)), W2b, b2b)
    assert _same_bits(ops.encode(*(t.to(dev()) for t in (x5, W1b, b1b, W2b, b2b))), want)


@pytest.mark.gpu
@pytest.mark.parametrize("hidden,n,e", [(64, 500, 7001), (128, 300, 2500), (64, 10, 3), (256, 300, 2501)])   # 256: round 4
def test_edge_gate_ref_equals_the_reference_sequence(hidden, n, e):
    from gnnome_amd import ops
    g = torch.Generator().manual_seed(hidden + e)
    src = torch.randint(0, n, (e,), generator=g).int()
    dst = torch.randint(0, n, (e,), generator=g).int()
    gv, cv = ops.GraphViews(src.to(dev()), dst.to(dev()), n), cpu_ops.CpuViews(src, dst, n)
    P = torch.randn(n, 2 * hidden, generator=g)
    W3, b3 = torch.randn(hidden, hidden, generator=g) / hidden ** 0.5, torch.randn(hidden, generator=g)
    scale, shift = 40 * torch.rand(hidden, generator=g), torch.randn(hidden, generator=g)
    e0 = 30 * torch.randn(e, hidden, generator=g)
    enc = (torch.randn(16, 2, generator=g), torch.randn(16, generator=g), torch.randn(hidden, 16, generator=g), torch.randn(hidden, generator=g))
    e_raw = torch.randn(e, 2, generator=g)
    d = lambda t: t.to(dev())  # noqa: E731
    Pd = d(P)

    def expect(e_rows):   # gated_gcn_full.py:97,104-110 with every nn.Linear as the k-ordered chain
        s_, d_ = cv.srt_src.long(), cv.srt_dst.long()
        x = (P[:, :hidden][s_] + P[:, hidden:][d_]) + _chain(e_rows, W3, b3)
        return torch.relu(cpu_ops._fma(x, scale, shift)) + e_rows

    got = ops.edge_gate_ref(d(e0), Pd[:, :hidden], Pd[:, hidden:], gv, d(W3), d(b3), d(scale), d(shift))
    assert _same_bits(got, expect(e0))
    enc_rows = _chain(torch.relu(_chain(e_raw[cv.srt_eid.long()], enc[0], enc[1])), enc[2], enc[3])
    got = ops.edge_gate_ref(None, Pd[:, :hidden], Pd[:, hidden:], gv, d(W3), d(b3), d(scale), d(shift),
                            raw_edges=(d(e_raw), tuple(d(t) for t in enc)))
    assert _same_bits(got, expect(enc_rows))


@pytest.mark.gpu
@pytest.mark.parametrize("hidden,e", [(64, 70_001), (128, 33_333), (64, 31)])
def test_matrix_core_and_valu_reference_order_kernels_give_the_same_bits(hidden, e):
    """Round 3: the chain runs on the fp32 matrix cores (v_mfma_f32_32x32x2_f32 adds its two k terms as an fma chain, k0 first:
    tools/mfma_order_probe.hip); round 2's scalar-fed VALU kernels are the same function - every bit equal, in place too."""
    from gnnome_amd import ops
    g = torch.Generator().manual_seed(e)
    n = max(e // 9, 4)
    src, dst = torch.randint(0, n, (e,), generator=g).int(), torch.randint(0, n, (e,), generator=g).int()
    gv = ops.GraphViews(src.to(dev()), dst.to(dev()), n)
    d = lambda t: t.to(dev())  # noqa: E731
    A, W, b = 10 * torch.randn(n, hidden, generator=g), 0.3 * torch.randn(5 * hidden, hidden, generator=g), torch.randn(5 * hidden, generator=g)
    W3, b3 = torch.randn(hidden, hidden, generator=g) / hidden ** 0.5, torch.randn(hidden, generator=g)
    scale, shift = 40 * torch.rand(hidden, generator=g), torch.randn(hidden, generator=g)
    e0, e_raw = 30 * torch.randn(e, hidden, generator=g), torch.randn(e, 2, generator=g)
    enc = tuple(d(t) for t in (torch.randn(16, 2, generator=g), torch.randn(16, generator=g), torch.randn(hidden, 16, generator=g),
                               torch.randn(hidden, generator=g)))
    out = {}
    try:
        for variant in (0, 1):
            ops.set_tuning(8, variant)
            P = ops.linear_ref(d(A), d(W), d(b))
            in_place = d(e0).clone()
            ops.edge_gate_ref(in_place, P[:, 3 * hidden:4 * hidden], P[:, 4 * hidden:], gv, d(W3), d(b3), d(scale), d(shift))
            fresh = ops.edge_gate_ref(None, P[:, 3 * hidden:4 * hidden], P[:, 4 * hidden:], gv, d(W3), d(b3), d(scale), d(shift),
                                      raw_edges=(d(e_raw), enc))
            out[variant] = (P.cpu(), in_place.cpu(), fresh.cpu())
    finally:
        ops.set_tuning(8, 0)
    for a, c in zip(out[0], out[1]):
        assert torch.equal(a, c)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["banded", "uniform"])
def test_ecoli_sized_graph_shipped_weights(shipped_weights, kind):
    """SURVEY.md 8d's substitute for BASELINE configs[0]: the one configuration where two correct fp32 evaluations of
    the model differ by more than 1e-4 (layer 0's bn_e gain of 135) - the HIP path must land on the REFERENCE's one."""
    from oracle import c_oracle
    n, e = ECOLI
    gr = make_graph(n, e, seed=1, kind=kind)
    x = degree_features(gr["src"], gr["dst"], n)
    graph = (gr["src"], gr["dst"], n)
    with torch.no_grad():
        ref32 = model_from_state_dict(shipped_weights).eval()(graph, x, gr["e"])
    ref64 = c_oracle.forward(shipped_weights, gr["src"], gr["dst"], n, x, gr["e"], precision="f64")
    m = gnnome_amd.SymGatedGCNModel(2, 2, 64, 16, 8, 64, "batch").eval()
    m.load_state_dict(shipped_weights)
    m.to(dev())
    res = {"oracle32_vs_fp64": _dprob(ref32, ref64)}
    for mode in ("auto", "reference", "fast"):
        m.arithmetic = mode
        out = m(graph, x.to(dev()), gr["e"].to(dev()))
        res[mode + "_vs_oracle32"] = _dprob(out, ref32)
        res[mode + "_vs_fp64"] = _dprob(out, ref64)
    _record("ecoli_sized_shipped_weights", kind=kind, nodes=n, edges=e, **res)
    assert res["auto_vs_oracle32"] < PROB_TOL and res["reference_vs_oracle32"] < PROB_TOL, res
    assert res["auto_vs_oracle32"] < AUTO_TOL, res
    # the distance to the truth is the model's own fp32 noise, whichever fp32 evaluation is taken
    assert res["auto_vs_fp64"] < 3 * max(res["oracle32_vs_fp64"], 5e-5), res
    assert res["fast_vs_fp64"] < 3 * max(res["oracle32_vs_fp64"], 5e-5), res


@pytest.mark.gpu
def test_high_gain_layer_at_hidden_256():
    """VERDICT r3 missing item 5 / next-round item 8: no H = 256 checkpoint exists, so a synthetic one stands in - random-init weights whose
    layer-0 bn_e has channels with running_var ~ 5e-5, i.e. a gain gamma / sqrt(var + eps) of ~130 like the shipped H = 64 checkpoint's.
    "auto" must send that layer (and only that one) through the reference-order kernels and land on the REFERENCE's fp32 result (the torch-CPU
    oracle) far inside the 1e-4 bar; "reference" likewise; and every mode stays within the model's own fp32 noise of the fp64 truth."""
    from gnnome_amd.synth import random_state_dict
    from gnnome_amd import engine
    hidden, n, e = 256, 6000, 60_000
    sd = random_state_dict(hidden, seed=4)
    g = torch.Generator().manual_seed(9)
    rv = sd["gnn.convs.0.bn_e.running_var"].clone()
    hot = torch.randperm(hidden, generator=g)[:24]
    rv[hot] = 5e-5 * (0.5 + torch.rand(24, generator=g))
    sd["gnn.convs.0.bn_e.running_var"] = rv
    sd["gnn.convs.0.bn_e.weight"] = sd["gnn.convs.0.bn_e.weight"].abs() + 0.5
    gr = make_graph(n, e, seed=2, kind="banded")
    x = degree_features(gr["src"], gr["dst"], n)
    graph = (gr["src"], gr["dst"], n)
    with torch.no_grad():
        ref32 = model_from_state_dict(sd).eval()(graph, x, gr["e"])
        ref64 = model_from_state_dict(sd, dtype=torch.float64).eval()(graph, x.double(), gr["e"].double()).float()
    m = gnnome_amd.SymGatedGCNModel(2, 2, hidden, 16, 8, 64, "batch").eval()
    m.load_state_dict(sd)
    m.to(dev())
    res = {"oracle32_vs_fp64": _dprob(ref32, ref64)}
    for mode in ("auto", "reference", "fast"):
        m.arithmetic = mode
        out = m(graph, x.to(dev()), gr["e"].to(dev()))
        res[mode + "_vs_oracle32"] = _dprob(out, ref32)
        res[mode + "_vs_fp64"] = _dprob(out, ref64)
        if mode == "auto":   # layer 0, and only layer 0, runs in the reference's order
            assert [lw.ref for lw in engine.Prepared(m, dev()).layers] == [True] + [False] * 7
    _record("high_gain_layer_h256", nodes=n, edges=e, **res)
    assert res["auto_vs_oracle32"] < AUTO_TOL and res["reference_vs_oracle32"] < AUTO_TOL, res
    assert all(res[k + "_vs_fp64"] < 3 * max(res["oracle32_vs_fp64"], 5e-5) for k in ("auto", "reference", "fast")), res


def _pin_order(conv):
    """Replace the dense layers and the eval BatchNorms of ONE oracle layer by the order csrc/reference_order*.hip implements: every output
    element one k-ascending chain of fused multiply-adds from zero, bias afterwards; BatchNorm as fma(x, alpha, fma(-mean, alpha, beta)).
    That IS what torch's CPU nn.Linear / BatchNorm1d compute on Intel hosts at these widths (tools/cpu_linear_order_probe256.py,
    profiles/r05_cpu_linear_order_k256.txt); the GPU pool's EPYC hosts follow it up to K = 192 and cut K = 256 into two blocks, and
    round the folded BatchNorm shift twice - so at H = 256 "the torch-CPU oracle" is itself host-dependent at the 1e-3 level on a layer
    as sensitive as the one below, and the order has to be pinned for the comparison to mean anything."""
    import types

    def chain_linear(self, a):
        acc = torch.zeros(a.shape[0], self.weight.shape[0], dtype=torch.float64)
        for k in range(a.shape[1]):     # fma: the product of two floats is exact in double, one rounding per step
            acc = (acc + a[:, k:k + 1].double() * self.weight[:, k].double().unsqueeze(0)).float().double()
        return (acc + self.bias.double()).float()

    def fma_bn(self, x):
        alpha = self.weight * (1.0 / torch.sqrt(self.running_var + self.eps))
        beta = (self.bias.double() - self.running_mean.double() * alpha.double()).float()
        return (x.double() * alpha.double() + beta.double()).float()
    for name in ("A_1", "A_2", "A_3", "B_1", "B_2", "B_3"):
        lin = getattr(conv, name)
        lin.forward = types.MethodType(chain_linear, lin)
    for name in ("bn_e", "bn_h"):
        bn = getattr(conv, name)
        bn.forward = types.MethodType(fma_bn, bn)


@pytest.mark.gpu
def test_reference_order_is_needed_and_sufficient_at_hidden_256():
    """the test above cannot tell the modes apart (2e-6 in all three).  Here layer 0's gate input is built the way
    the shipped checkpoint's is, only more so: a large common offset (B_3's bias ~ 300) with little variation (weights x 0.01), and a
    BatchNorm whose running statistics MATCH that input (a trained-like state: mean ~ 300, var ~ 5e-6, gain ~ 500) - its output is
    of order one and carries every bit of fp32 rounding of the pre-activation, magnified; the fp32 oracle is 1e-3 from the fp64 truth
    on it.  Against the oracle with layer 0 in the PINNED order (see _pin_order), "fast" (fp16x3 / bf16x6: another fp32 number) must
    MISS the 1e-4 bar and "auto" must send exactly that layer through the K = 256 reference-order kernels and hold it with a wide
    margin: the mechanism is needed, and it is sufficient.  What the host's own torch gives is recorded next to it (equal to the pinned
    oracle on Intel hosts; another fp32 evaluation on the pool's EPYC hosts, measured 9e-4 away)."""
    from gnnome_amd.synth import random_state_dict
    from gnnome_amd import engine
    hidden, n, e = 256, 6000, 60_000
    gr = make_graph(n, e, seed=2, kind="banded")
    x = degree_features(gr["src"], gr["dst"], n)
    graph = (gr["src"], gr["dst"], n)
    sd = random_state_dict(hidden, seed=4)
    for k in ("B_1", "B_2", "B_3"):
        sd[f"gnn.convs.0.{k}.weight"] = sd[f"gnn.convs.0.{k}.weight"] * 0.01
    sd["gnn.convs.0.B_3.bias"] = 300.0 * (1 + 0.3 * torch.randn(hidden, generator=torch.Generator().manual_seed(1)))
    threads = torch.get_num_threads()
    torch.set_num_threads(min(threads, 16))     # (thousands of small CPU ops below: a 256-thread pool makes them crawl)
    try:
        om = model_from_state_dict(sd, dtype=torch.float64).eval()
        with torch.no_grad():   # the statistics of layer 0's gate input on this graph, in fp64
            h = om.linear2_node(torch.relu(om.linear1_node(x.double())))
            ee = om.linear2_edge(torch.relu(om.linear1_edge(gr["e"].double())))
            c = om.gnn.convs[0]
            pre = c.B_1(h)[gr["src"].long()] + c.B_2(h)[gr["dst"].long()] + c.B_3(ee)
        sd["gnn.convs.0.bn_e.running_mean"] = pre.mean(0).float()
        sd["gnn.convs.0.bn_e.running_var"] = pre.var(0, unbiased=False).float()
        sd["gnn.convs.0.bn_e.weight"] = sd["gnn.convs.0.bn_e.weight"].abs() + 0.5
        pinned = model_from_state_dict(sd).eval()
        _pin_order(pinned.gnn.convs[0])
        with torch.no_grad():
            host32 = model_from_state_dict(sd).eval()(graph, x, gr["e"])
            ref32 = pinned(graph, x, gr["e"])
            ref64 = model_from_state_dict(sd, dtype=torch.float64).eval()(graph, x.double(), gr["e"].double()).float()
    finally:
        torch.set_num_threads(threads)
    m = gnnome_amd.SymGatedGCNModel(2, 2, hidden, 16, 8, 64, "batch").eval()
    m.load_state_dict(sd)
    m.to(dev())
    res = {"pinned_oracle32_vs_fp64": _dprob(ref32, ref64), "host_torch_vs_pinned_oracle32": _dprob(host32, ref32),
           "gain": float(engine.Prepared(m, dev()).layers[0].gain_e)}
    for mode in ("auto", "reference", "fast"):
        m.arithmetic = mode
        out = m(graph, x.to(dev()), gr["e"].to(dev()))
        res[mode + "_vs_pinned_oracle32"] = _dprob(out, ref32)
        res[mode + "_vs_host_torch"] = _dprob(out, host32)
        if mode == "auto":
            assert [lw.ref for lw in engine.Prepared(m, dev()).layers] == [True] + [False] * 7
    _record("sensitive_layer_h256", nodes=n, edges=e, **res)
    assert res["pinned_oracle32_vs_fp64"] > 3e-4 and res["gain"] > 300, res     # the construction is as sensitive as intended
    assert res["fast_vs_pinned_oracle32"] > PROB_TOL, res                         # without the reference-order kernels the bar is missed
    assert res["auto_vs_pinned_oracle32"] < AUTO_TOL and res["reference_vs_pinned_oracle32"] < AUTO_TOL, res


@pytest.mark.gpu
def test_goldens_in_every_arithmetic_mode(shipped_weights):
    m = gnnome_amd.SymGatedGCNModel(2, 2, 64, 16, 8, 64, "batch").eval()
    m.load_state_dict(shipped_weights)
    m.to(dev())
    for name in ("g1_hand.pt", "g2_uniform_1k.pt"):
        g = load_golden(name)
        for mode in ("auto", "reference", "fast"):
            m.arithmetic = mode
            out = m((g["src"], g["dst"], g["num_nodes"]), g["x"].to(dev()), g["e"].to(dev()))
            d = _dprob(out, g["logits"])
            _record("golden_" + name, mode=mode, vs_reference_golden=d)
            assert d < (AUTO_TOL if mode != "fast" else PROB_TOL), (name, mode, d)
