"""BatchNorm element-wise kernels on bf16 rows, 16 bytes per lane (csrc/fused.hip: k_bn_apply_bf16x8, k_bn_bwd_apply_bf16x8,
k_bn_bwd_stats_bf16x8) against the 8-byte kernels they replace (SGF_EW8=0) and against fp64 of the same bf16 inputs.
(large/ours.py:36-40,83-93: BatchNorm + relu + residual of a GraphConv layer, forward and differentiated.)"""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def _ew8(v):
    from sgformer_amd import _lib
    os.environ["SGF_EW8"] = str(v)
    _lib.load().sgf_reload_env()


@pytest.fixture
def both_paths():
    yield
    os.environ.pop("SGF_EW8", None)
    from sgformer_amd import _lib
    _lib.load().sgf_reload_env()


def _inputs(cuda, n, d, pad, seed):
    g = torch.Generator(device=cuda).manual_seed(seed)

    def rows():                                             # rows of a wider buffer: leading dimension d + pad
        return (torch.randn(n, d + pad, device=cuda, generator=g) * 1.5 + 0.3).to(torch.bfloat16)[:, :d]

    x, gy, gy2, res = rows(), rows(), rows(), rows()
    mean = torch.randn(d, device=cuda, generator=g) * 0.2 + 0.3
    rstd = torch.rand(d, device=cuda, generator=g) + 0.5
    gamma = torch.rand(d, device=cuda, generator=g) + 0.5
    beta = torch.randn(d, device=cuda, generator=g) * 0.1
    return x, gy, gy2, res, mean, rstd, gamma, beta


def _ulp(t):
    """one bf16 unit in the last place at the magnitude of t (fp64)"""
    return torch.clamp(t.abs(), min=2.0 ** -126).log2().floor().exp2() * 2.0 ** -7


@pytest.mark.parametrize("n", [1, 33, 1531, 70001])
@pytest.mark.parametrize("d,pad", [(256, 0), (64, 0), (104, 0), (128, 8), (512, 0), (100, 0), (256, 4)])
@pytest.mark.parametrize("relu,use_res", [(True, True), (False, False), (True, False)])
def test_bn_apply_bf16_rows(cuda, both_paths, n, d, pad, relu, use_res):
    from sgformer_amd import ops
    K = ops.K
    x, gy, gy2, res, mean, rstd, gamma, beta = _inputs(cuda, n, d, pad, 7 * n + d)
    r = res if use_res else None
    _ew8(1)
    y8 = K.bn_apply(x, mean, rstd, gamma, beta, r, relu)
    _ew8(0)
    y4 = K.bn_apply(x, mean, rstd, gamma, beta, r, relu)
    assert torch.equal(y8, y4)                              # same arithmetic per element, same rounding
    ref = (x.double() - mean.double()) * rstd.double() * gamma.double() + beta.double()
    if relu:
        ref = torch.relu(ref)
    if use_res:
        ref = ref + res.double()
    err = (y8.double() - ref).abs()
    assert bool((err <= 0.51 * _ulp(ref) + 1e-5).all())      # the fp32 result (|z| <= ~15: 1e-5) rounded ONCE


@pytest.mark.parametrize("n", [1, 33, 1531, 70001])
@pytest.mark.parametrize("d,pad", [(256, 0), (64, 0), (104, 0), (128, 8), (512, 0), (100, 0), (256, 4)])
@pytest.mark.parametrize("relu,training,two", [(True, True, False), (True, True, True), (False, True, False), (True, False, False)])
def test_bn_backward_bf16_rows(cuda, both_paths, n, d, pad, relu, training, two):
    from sgformer_amd import ops
    K = ops.K
    x, gy, gy2, res, mean, rstd, gamma, beta = _inputs(cuda, n, d, pad, 3 * n + d)
    out = {}
    for v in (1, 0):
        _ew8(v)
        stats = K.bn_bwd_stats2(gy, gy2, x, mean, rstd, gamma, beta, relu) if two else K.bn_bwd_stats(gy, x, mean, rstd, gamma, beta, relu)
        out[v] = stats
    # fp64 of the same inputs
    xh = (x.double() - mean.double()) * rstd.double()
    g = gy.double() + (gy2.double() if two else 0.0)
    if relu:
        # the mask is the FORWARD's decision: fp32 fma(xh, gamma, beta) > 0 (elements within rounding of 0 may differ from fp64)
        z32 = torch.addcmul(beta, ((x.float() - mean) * rstd), gamma)
        sure = (xh * gamma.double() + beta.double()).abs() > 1e-5
        g = torch.where(z32 > 0, g, torch.zeros_like(g))
    else:
        sure = torch.ones_like(xh, dtype=torch.bool)
    s0, s1 = g.sum(0), (g * xh).sum(0)
    scale0 = g.abs().sum(0) + 1e-30
    scale1 = (g * xh).abs().sum(0) + 1e-30
    for v in (1, 0):
        st = out[v].double()
        slack = ((~sure).double() * (gy.double().abs() + 1)).sum(0) * 4        # borderline mask elements, if any
        assert bool(((st[:d] - s0).abs() <= 2e-6 * scale0 + slack).all()), v
        assert bool(((st[d:] - s1).abs() <= 2e-6 * scale1 + slack * 4).all()), v
    if two:
        return
    # dx with the SAME statistics on both paths: bit for bit
    stats = out[1]
    _ew8(1)
    dx8 = K.bn_bwd_apply(gy, x, mean, rstd, gamma, beta, relu, stats, 1.0 / max(n, 1), training)
    _ew8(0)
    dx4 = K.bn_bwd_apply(gy, x, mean, rstd, gamma, beta, relu, stats, 1.0 / max(n, 1), training)
    assert torch.equal(dx8, dx4)
    ge = gy.double()
    if relu:
        ge = torch.where(z32 > 0, ge, torch.zeros_like(ge))
    if training:
        ge = ge - (stats[:d].double() / max(n, 1) + xh * stats[d:].double() / max(n, 1))
    ref = gamma.double() * rstd.double() * ge
    err = (dx8.double() - ref).abs()
    assert bool((err <= 0.51 * _ulp(ref) + 1e-5).all())


@pytest.mark.parametrize("n", [1, 33, 1531, 70001])
@pytest.mark.parametrize("d,pad", [(256, 0), (64, 0), (128, 8), (512, 0), (100, 0), (256, 4)])
@pytest.mark.parametrize("relu,use_ln,use_res,a,b", [(True, True, True, 0.5, 0.5), (False, True, True, 0.3, 0.7),
                                                    (True, True, False, 1.0, 0.0), (True, False, True, 0.5, 0.5)])
def test_ln_backward_bf16_rows(cuda, both_paths, n, d, pad, relu, use_ln, use_res, a, b):
    """k_ln_bwd_bf16x8 (large/ours.py:205-211 differentiated: residual mix, LayerNorm, relu) against the 8-byte kernel and
    against fp64 autograd of the same bf16 inputs."""
    from sgformer_amd import ops
    K = ops.K
    x, gy, _, res, *_ = _inputs(cuda, n, d, pad, 11 * n + d)
    g = torch.Generator(device=cuda).manual_seed(d)
    gamma = (torch.rand(d, device=cuda, generator=g) + 0.5) if use_ln else None
    beta = (torch.randn(d, device=cuda, generator=g) * 0.1) if use_ln else None
    r = res if use_res else None
    _ew8(1)
    y, mean, rstd = K.ln_fwd(x, r, a, b, gamma, beta, relu, 1e-5)
    out = {}
    for v in (1, 0):
        _ew8(v)
        out[v] = K.ln_bwd(gy, y, x, r, a, b, gamma, relu, mean, rstd)
    # fp64 autograd of the same computation (mask from the forward's stored y, as the kernels take it)
    xd = x.double().requires_grad_(True)
    rd = res.double().requires_grad_(True)
    pre = a * xd + (b * rd if use_res else 0.0)
    gd = gamma.double().requires_grad_(True) if use_ln else None
    bd = beta.double().requires_grad_(True) if use_ln else None
    z = torch.nn.functional.layer_norm(pre, (d,), gd, bd, 1e-5) if use_ln else pre
    gyd = gy.double()
    if relu:
        gyd = torch.where(y.double() > 0, gyd, torch.zeros_like(gyd))
    z.backward(gyd)
    for v in (1, 0):
        dx, dres, dgamma, dbeta = out[v]
        tol = 0.51 * _ulp(xd.grad) + 3e-5 * (gy.double().abs().amax() + 1)
        assert bool(((dx.double() - xd.grad).abs() <= tol).all()), v
        if use_res:
            tol = 0.51 * _ulp(rd.grad) + 3e-5 * (gy.double().abs().amax() + 1)
            assert bool(((dres.double() - rd.grad).abs() <= tol).all()), v
        if use_ln:
            scale = gyd.abs().sum(0) + 1
            assert bool(((dbeta.double() - bd.grad).abs() <= 2e-6 * scale).all()), v
            assert bool(((dgamma.double() - gd.grad).abs() <= 3e-5 * scale * 4).all()), v
    # the two kernels differ in the order of a row's sums only: one bf16 step at most (values near zero: the fp32 noise), rarely
    d8, d4 = out[1][0].double(), out[0][0].double()
    assert bool(((d8 - d4).abs() <= 1.01 * _ulp(torch.maximum(d8.abs(), d4.abs())) + 1e-4).all())
    assert float((d8 != d4).double().mean()) <= 0.02
