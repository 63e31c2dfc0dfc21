"""gnnome_linear_planes_f32 / gnnome_weight_planes_f16 (csrc/node_project.hip, round 6): the node projection
P = h Wcat^T + bcat (gated_gcn_full.py:91-96 as one GEMM) and the scorer's node halves (score_predictor.py:13-14) on the
fp16x3 planes of the weights.

Checked against an fp64 product at the kernel bar (1e-5 of scale), against a numpy restatement of the fp16x3 arithmetic itself
(same planes, same three products: what is left is fp32 summation order), and through the properties the engine relies on: a row's
bits do not depend on the row count or on the launch, strided inputs / outputs leave their neighbours alone, an operand outside
fp16's range makes ITS row NaN and no other.
"""
import numpy as np
import pytest
import torch

from gnnome_amd import ops

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda", 0)


def _close(got, want64, scale, tol=1e-5):
    err = (got.double().cpu() - want64).abs().max().item()
    assert err <= tol * max(scale, 1.0), f"max abs err {err:.3e} vs scale {scale:.3e}"


def _planes_np(x):
    """x = x1 + x2 / 2048 (+ <= 2^-22 |x|): the two fp16 planes of edge_tile_f16.hip's header."""
    x = x.astype(np.float32)
    x1 = x.astype(np.float16)
    x2 = ((x - x1.astype(np.float32)) * np.float32(2048.0)).astype(np.float16)
    return x1.astype(np.float64), x2.astype(np.float64)


@pytest.mark.parametrize("m,k,nout", [(1, 128, 640), (31, 128, 640), (128, 128, 640), (129, 128, 640), (333, 128, 256), (5000, 128, 640),
                                      (100_003, 128, 640), (7, 256, 1280), (129, 256, 1280), (4097, 256, 1280), (60_001, 256, 1280),
                                      (1000, 128, 128), (1000, 256, 128), (257, 128, 64), (300, 256, 64), (513, 128, 1536),
                                      (1, 64, 320), (127, 64, 320), (129, 64, 320), (4099, 64, 320), (100_003, 64, 320), (999, 64, 128), (1500, 64, 64)])
def test_linear_planes_against_fp64(m, k, nout):
    g = torch.Generator().manual_seed(m + k + nout)
    A, W, b = torch.randn(m, k, generator=g), torch.randn(nout, k, generator=g) / k ** 0.5, torch.randn(nout, generator=g)
    want = A.double() @ W.double().t() + b.double()
    Ad, Wd, bd = A.to(dev()), W.to(dev()), b.to(dev())
    planes = ops.weight_planes(Wd)
    assert planes.numel() * 2 == nout * k * 4
    got = ops.linear(Ad, Wd, bd, planes=planes)
    _close(got, want, scale=4.0)
    # the arithmetic itself: the three plane products in fp64 leave only the fp32 accumulation (K / 16 roundings of partial sums of 16)
    a1, a2 = _planes_np(A.numpy())
    w1, w2 = _planes_np(W.numpy())
    model = a1 @ w1.T + (a1 @ w2.T + a2 @ w1.T) / 2048.0 + b.double().numpy()
    assert np.abs(got.double().cpu().numpy() - model).max() <= 4e-6
    # no bias
    _close(ops.linear(Ad, Wd, None, planes=planes), want - b.double(), scale=4.0)
    if nout % 128 == 0 and (k == 256 or nout >= 256):   # the shapes ops.linear routes here by itself
        assert torch.equal(ops.linear(Ad, Wd, bd), got)


@pytest.mark.parametrize("k,nout", [(64, 320), (128, 640), (256, 1280)])
def test_linear_planes_rows_do_not_depend_on_the_launch(k, nout):
    """engine / dist cut the node range freely (owned rows, halo rows, pipeline chunks): a row's bits are a function of the row."""
    g = torch.Generator().manual_seed(k)
    m = 3000
    A = torch.randn(m, k, generator=g).to(dev())
    W = (torch.randn(nout, k, generator=g) / k ** 0.5).to(dev())
    b = torch.randn(nout, generator=g).to(dev())
    planes = ops.weight_planes(W)
    whole = ops.linear(A, W, b, planes=planes)
    for lo, hi in ((0, 1), (0, 127), (5, 133), (1000, 3000), (2999, 3000)):
        assert torch.equal(ops.linear(A[lo:hi], W, b, planes=planes), whole[lo:hi]), (lo, hi)
    again = [ops.linear(A, W, b, planes=planes) for _ in range(20)]
    assert all(torch.equal(x, whole) for x in again)


def test_linear_planes_strided_views_and_selector():
    g = torch.Generator().manual_seed(11)
    m, k, nout = 700, 128, 640
    wide_in = torch.randn(m, k + 64, generator=g).to(dev())
    A = wide_in[:, 32:32 + k]                      # row stride k + 64, 16-byte aligned
    Wfull = torch.randn(nout, 3 * k, generator=g).to(dev())
    W = Wfull[:, k:2 * k]                          # a column block of a wider matrix (predictor.W1's node halves are such views)
    planes = ops.weight_planes(W)
    wide = torch.full((m, nout + 128), 7.0, device=dev())
    ops.linear(A, W, None, out=wide[:, 64:64 + nout], planes=planes)
    want = A.double().cpu() @ W.double().cpu().t()
    _close(wide[:, 64:64 + nout], want, scale=float(k) ** 0.5 * 4)
    assert (wide[:, :64] == 7.0).all() and (wide[:, 64 + nout:] == 7.0).all()
    # a selector matrix picks columns of W: 1.0 * (w1 + w2 / 2048) is w to 2^-22 - and detects any transposition / permutation of k
    sel = torch.zeros(m, k)
    sel[torch.arange(m), torch.arange(m) % k] = 1.0
    got = ops.linear(sel.to(dev()), W.contiguous(), None, planes=planes)
    assert torch.allclose(got.cpu(), W.cpu().t()[torch.arange(m) % k], rtol=2.0 ** -21, atol=1e-9)


@pytest.mark.parametrize("k,nout", [(64, 320), (128, 640), (256, 1280)])
def test_linear_planes_out_of_range_row_is_loud(k, nout):
    g = torch.Generator().manual_seed(3)
    A = torch.randn(300, k, generator=g)
    A[17, 5] = 7.0e4       # beyond fp16
    A[200, k - 1] = float("nan")
    W = torch.randn(nout, k, generator=g) / k ** 0.5
    Wd = W.to(dev())
    got = ops.linear(A.to(dev()), Wd, None, planes=ops.weight_planes(Wd)).cpu()
    bad = ~torch.isfinite(got).all(1)
    assert bad[17] and bad[200] and int(bad.sum()) == 2
    good = ~bad
    _close(got[good], A[good].double() @ W.double().t(), scale=4.0)


def test_linear_planes_rejects_what_it_is_not_built_for():
    A = torch.randn(10, 32, device=dev())
    W = torch.randn(640, 32, device=dev())
    with pytest.raises(Exception):
        ops.weight_planes(W)            # K = 32
    W = torch.randn(100, 128, device=dev())
    with pytest.raises(Exception):
        ops.weight_planes(W)            # Nout % 32
    # the bf16x6 arithmetic switch keeps its own kernels (fp32's range)
    A = torch.randn(300, 128, device=dev())
    A[3, 3] = 1.0e5
    W = torch.randn(640, 128, device=dev()) / 11.0
    with ops.bf16x6_arithmetic():
        assert torch.isfinite(ops.linear(A, W, None)).all()
