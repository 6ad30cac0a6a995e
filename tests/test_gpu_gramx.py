"""csrc/gramx.hip — the LDS-DMA node reductions behind sgf_gram / sgf_gram2 / sgf_attn_h_bwd_reduce_scaled (bf16, n >= 4096,
16-byte aligned rows): against fp64 of the same bf16 operands, against the register-staged kernels they replace
(SGF_GRAMX=0), on ragged row counts (the cleared tail of the last stage), strided operands (a column slice of a wider
buffer), narrow operands (the head's dW: m = 48; clamped image columns) and every slot of the 4-stage ring.

Reference lines: large/ours.py:36-40,:77,:198,:275 differentiated (dW = g^T x, db = sum g) and :130-151 (attention
backward, SURVEY.md Appendix B: dS = sum_n q_n^T dnum_n, dz = sum_n q_n dden_n)."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-300))


def _switch(on: bool):
    from sgformer_amd import _lib
    os.environ["SGF_GRAMX"] = "1" if on else "0"
    _lib.load().sgf_reload_env()


def _bn2(on: bool):
    """sgf_gram2_bn_bwd is opt-in (SGF_GRAM_BN2=1: measured level with the two kernels it replaces)"""
    from sgformer_amd import _lib
    os.environ["SGF_GRAM_BN2"] = "1" if on else "0"
    _lib.load().sgf_reload_env()


@pytest.fixture(autouse=True)
def _restore():
    yield
    os.environ.pop("SGF_GRAMX", None)
    os.environ.pop("SGF_GRAM_BN2", None)
    from sgformer_amd import _lib
    _lib.load().sgf_reload_env()


# n: one stage per block and fewer blocks than CUs; exactly the ring depth per block; ragged tails of 1 / 31 rows; several
# laps of the ring
@pytest.mark.parametrize("n", [4096, 4097, 4127, 8192 + 5, 32 * 256 * 4, 32 * 256 * 4 + 17, 100003, 300000])
@pytest.mark.parametrize("m,k", [(256, 256), (48, 256), (256, 104), (128, 128), (64, 200), (8, 8), (200, 56)])
def test_gramx_matches_fp64_and_old_kernel(cuda, n, m, k):
    from sgformer_amd import ops
    if n > 100003 and (m, k) not in ((256, 256), (48, 256)):
        pytest.skip("large n: the production shapes only")
    K = ops.K
    g = torch.Generator().manual_seed(n + 7 * m + k)
    a = (torch.randn(n, m, generator=g) * 0.5 + 0.3).bfloat16().to(cuda)
    b = (torch.randn(n, k, generator=g) * 1.5 - 0.2).bfloat16().to(cuda)
    _switch(True)
    c, cs = K.gram(a, b)
    c2, cs2 = K.gram(a, b)
    assert torch.equal(c, c2) and torch.equal(cs, cs2)              # deterministic
    ad, bd = a.double(), b.double()
    ref = ad.t() @ bd
    assert _rel(c, ref) <= 5e-6, _rel(c, ref)
    assert _rel(cs, ad.sum(0)) <= 5e-6
    _switch(False)
    c_old, cs_old = K.gram(a, b)
    assert _rel(c, c_old) <= 2e-6 and _rel(cs, cs_old) <= 2e-6      # only the summation order differs


def test_gramx_is_transpose_aware(cuda):
    """asymmetric operands: C = A^T B and not B^T A / a row-column swap of the fragments (cdna_hip_programming.md §3)."""
    from sgformer_amd import ops
    n, m, k = 8192, 256, 256
    a = torch.zeros(n, m)
    b = torch.zeros(n, k)
    rows = torch.arange(n)
    a[rows, rows % m] = 1.0                                         # A^T B [i, j] = sum over rows with row % m == i of B[row, j]
    b[:] = (torch.arange(k).float() * 0.5 + 1.0)[None, :] * ((rows % 7).float() + 1.0)[:, None]
    _switch(True)
    c, cs = ops.K.gram(a.bfloat16().to(cuda), b.bfloat16().to(cuda))
    ref = a.bfloat16().double().t() @ b.bfloat16().double()
    assert _rel(c, ref) <= 1e-6
    assert torch.equal(cs.cpu(), a.sum(0))


@pytest.mark.parametrize("n", [5000, 65536 + 9])
def test_gramx_strided_operands_and_sliced_output(cuda, n):
    """operands that are column slices of a wider buffer ([Q | K | V] style: ld = 3 d), output into a column slice."""
    from sgformer_amd import ops
    g = torch.Generator().manual_seed(n)
    wide = torch.randn(n, 768, generator=g).bfloat16().to(cuda)
    a, b = wide[:, 256:512], wide[:, 512:768]
    _switch(True)
    big = torch.full((256, 256 + 12), 7.0, device=cuda)
    ops.K.gram(a, b, out=big[:, 4:260], want_colsum=False)
    assert _rel(big[:, 4:260], a.double().t() @ b.double()) <= 5e-6
    assert bool((big[:, :4] == 7.0).all()) and bool((big[:, 260:] == 7.0).all())


@pytest.mark.parametrize("n,m,k", [(4099, 128, 128), (50001, 256, 256), (20000, 48, 256), (262144 + 33, 256, 256)])
def test_gramx_paired(cuda, n, m, k):
    from sgformer_amd import ops
    K = ops.K
    g = torch.Generator().manual_seed(n + m)
    a = torch.randn(n, m, generator=g).bfloat16().to(cuda)
    b1 = torch.randn(n, k, generator=g).bfloat16().to(cuda)
    b2 = (torch.randn(n, k, generator=g) + 0.5).bfloat16().to(cuda)
    _switch(True)
    dw = torch.empty(m, 2 * k, device=cuda)
    cs = K.gram2(a, b1, b2, dw[:, :k], dw[:, k:], want_colsum=True)
    ad = a.double()
    assert _rel(dw[:, :k], ad.t() @ b1.double()) <= 2e-6
    assert _rel(dw[:, k:], ad.t() @ b2.double()) <= 2e-6
    assert _rel(cs, ad.sum(0)) <= 2e-6 or float(ad.sum(0).abs().max()) < 1e-2
    dw2 = torch.empty_like(dw)
    cs2 = K.gram2(a, b1, b2, dw2[:, :k], dw2[:, k:], want_colsum=True)
    assert torch.equal(dw, dw2) and torch.equal(cs, cs2)
    _switch(False)
    dw3 = torch.empty_like(dw)
    K.gram2(a, b1, b2, dw3[:, :k], dw3[:, k:], want_colsum=False)
    assert _rel(dw, dw3) <= 2e-6


@pytest.mark.parametrize("n,d", [(4096, 256), (4097, 64), (20001, 256), (100003, 128), (300007, 256)])
def test_gramx_attention_backward_reduce(cuda, n, d):
    """sgf_attn_h_bwd_reduce_scaled on the DMA kernel: hstats = [h^T dnum | sum h dden | sum dnum | sum dden] with
    dnum = g / den re-rounded to bf16 for the matrix cores (2e-3 on the matrix, as the register-staged kernel) and the
    vectors in fp32 (1e-5); identical run to run; and against the old kernel."""
    from sgformer_amd import ops
    K = ops.K
    g_ = torch.Generator().manual_seed(n + d)
    h = torch.randn(n, d, generator=g_).bfloat16().to(cuda)
    g = torch.randn(n, d, generator=g_).bfloat16().to(cuda)
    inv = (1.0 / (1.0 + torch.rand(n, generator=g_))).to(cuda)
    dden = torch.randn(n, generator=g_).to(cuda)
    rowscal = torch.stack([inv, dden], 1).contiguous()
    _switch(True)
    hs = K.attn_h_bwd_reduce_scaled(h, g, rowscal)
    hs2 = K.attn_h_bwd_reduce_scaled(h, g, rowscal)
    assert torch.equal(hs, hs2)
    hd, gd = h.double(), g.double()
    dnum = gd * inv.double()[:, None]
    assert _rel(hs[:d * d].reshape(d, d), hd.t() @ dnum) <= 2e-3
    assert _rel(hs[d * d:d * d + d], (hd * dden.double()[:, None]).sum(0)) <= 1e-5
    assert _rel(hs[d * d + d:d * d + 2 * d], dnum.sum(0)) <= 1e-5
    assert abs(float(hs[-1]) - float(dden.double().sum())) <= 1e-5 * max(1.0, float(dden.double().abs().sum()))
    # the matrix against fp64 of the ROUNDED dnum: fp32 sums of exact products
    dnum_r = (g.float() * inv[:, None]).bfloat16().double()
    assert _rel(hs[:d * d].reshape(d, d), hd.t() @ dnum_r) <= 5e-6
    _switch(False)
    hs_old = K.attn_h_bwd_reduce_scaled(h, g, rowscal)
    assert _rel(hs, hs_old) <= 2e-6


# ---- k_gramt: the stems' dW / db with the A operand formed in LDS (sgf_gram_bn_bwd / sgf_gram_ln_bwd) ---------------------
@pytest.mark.parametrize("n", [4096, 4099, 8192 + 31, 65536 + 7, 300001])
@pytest.mark.parametrize("m,k", [(256, 104), (256, 128), (128, 104), (64, 64), (256, 8), (200, 72)])
@pytest.mark.parametrize("two,relu,training", [(True, True, True), (False, True, True), (True, False, False)])
def test_gramt_bn(cuda, n, m, k, two, relu, training):
    """large/ours.py:77-80 differentiated (GraphConv's stem: Linear -> BatchNorm -> relu): dW = dz^T x, db = sum dz with
    dz = BatchNorm'(relu'(g1 + g2)) never written.  fp64 on the same bf16 operands, dz rounded to bf16 once; and the
    register-staged kernel on the same inputs."""
    from sgformer_amd import ops
    if n > 65536 + 7 and (m, k) != (256, 104):
        pytest.skip("large n: the production shape only")
    K = ops.K
    g = torch.Generator().manual_seed(3 * n + m + k + two)
    g1 = torch.randn(n, m, generator=g).bfloat16().to(cuda)
    g2 = torch.randn(n, m, generator=g).bfloat16().to(cuda) if two else None
    z = (torch.randn(n, m, generator=g) * 1.3 + 0.2).bfloat16().to(cuda)
    x = torch.randn(n, k, generator=g).bfloat16().to(cuda)
    mean = (torch.randn(m, generator=g) * 0.2 + 0.2).to(cuda)
    rstd = (1.0 / (1.0 + torch.rand(m, generator=g))).to(cuda)
    gamma = (1.0 + 0.3 * torch.randn(m, generator=g)).to(cuda)
    beta = (0.2 * torch.randn(m, generator=g)).to(cuda)
    gsum = g1.float() + (g2.float() if two else 0.0)
    stats = K.bn_bwd_stats2(g1, g2, z, mean, rstd, gamma, beta, relu)
    inv_n = 1.0 / n
    _switch(True)
    dw, db = K.gram_bn_bwd(g1, g2, z, mean, rstd, gamma, beta, relu, stats, inv_n, training, x)
    dw2, db2 = K.gram_bn_bwd(g1, g2, z, mean, rstd, gamma, beta, relu, stats, inv_n, training, x)
    assert torch.equal(dw, dw2) and torch.equal(db, db2)
    xh = (z.double() - mean.double()) * rstd.double()
    gm = gsum.double() * ((xh * gamma.double() + beta.double()) > 0) if relu else gsum.double()
    corr = (stats[:m].double() * inv_n + xh * stats[m:].double() * inv_n) if training else 0.0
    dz_r = (gamma.double() * rstd.double() * (gm - corr)).float().bfloat16().double()
    ref = dz_r.t() @ x.double()
    assert _rel(dw, ref) <= 3e-4, _rel(dw, ref)
    assert _rel(db, dz_r.sum(0)) <= 3e-4 or float(dz_r.sum(0).abs().max()) < 1e-2
    _switch(False)
    dw_old, db_old = K.gram_bn_bwd(g1, g2, z, mean, rstd, gamma, beta, relu, stats, inv_n, training, x)
    assert _rel(dw, dw_old) <= 5e-5 and (_rel(db, db_old) <= 5e-5 or float(db_old.abs().max()) < 1e-2)


@pytest.mark.parametrize("n", [4096, 4099, 8192 + 31, 65536 + 7, 300001])
@pytest.mark.parametrize("m,k", [(256, 104), (256, 128), (128, 104), (64, 64), (256, 8)])
@pytest.mark.parametrize("relu,affine", [(True, True), (False, True), (True, False)])
def test_gramt_ln(cuda, n, m, k, relu, affine):
    """large/ours.py:198-201 differentiated (TransConv's stem: Linear -> LayerNorm -> relu): dW = dl^T x, db = sum dl,
    d gamma = sum g' xhat, d beta = sum g' with dl = LayerNorm'(relu'(g)) never written."""
    from sgformer_amd import ops
    if n > 65536 + 7 and (m, k) != (256, 104):
        pytest.skip("large n: the production shape only")
    K = ops.K
    g = torch.Generator().manual_seed(5 * n + m + k)
    gr = torch.randn(n, m, generator=g).bfloat16().to(cuda)
    xin = (torch.randn(n, m, generator=g) * 1.7 + 0.3).bfloat16().to(cuda)
    x = torch.randn(n, k, generator=g).bfloat16().to(cuda)
    gamma = (1.0 + 0.3 * torch.randn(m, generator=g)).to(cuda) if affine else None
    beta = (0.2 * torch.randn(m, generator=g)).to(cuda) if affine else None
    mean = xin.float().mean(1)
    rstd = (xin.float().var(1, unbiased=False) + 1e-5).rsqrt()
    _switch(True)
    dw, db, dg, dbt = K.gram_ln_bwd(gr, xin, mean, rstd, gamma, beta, relu, x)
    dw2, db2, dg2, dbt2 = K.gram_ln_bwd(gr, xin, mean, rstd, gamma, beta, relu, x)
    assert torch.equal(dw, dw2) and torch.equal(db, db2) and torch.equal(dg, dg2) and torch.equal(dbt, dbt2)
    xh = (xin.double() - mean.double()[:, None]) * rstd.double()[:, None]
    ga = gamma.double() if affine else 1.0
    be = beta.double() if affine else 0.0
    gm = gr.double() * ((xh * ga + be) > 0) if relu else gr.double()
    dxh = gm * ga
    dl = rstd.double()[:, None] * (dxh - dxh.mean(1, keepdim=True) - xh * (dxh * xh).mean(1, keepdim=True))
    dl_r = dl.float().bfloat16().double()
    ref = dl_r.t() @ x.double()
    assert _rel(dw, ref) <= 3e-4, _rel(dw, ref)
    assert _rel(db, dl_r.sum(0)) <= 3e-4 or float(dl_r.sum(0).abs().max()) < 1e-2
    assert bool(((dg.double() - (gm * xh).sum(0)).abs() <= 2e-6 * (gm * xh).abs().sum(0).clamp_min(1e-3)).all())
    assert bool(((dbt.double() - gm.sum(0)).abs() <= 2e-6 * gm.abs().sum(0).clamp_min(1e-3)).all())
    _switch(False)
    dw_old, db_old, dg_old, dbt_old = K.gram_ln_bwd(gr, xin, mean, rstd, gamma, beta, relu, x)
    assert _rel(dw, dw_old) <= 3e-4
    assert _rel(dg, dg_old) <= 1e-5 and _rel(dbt, dbt_old) <= 1e-5


# ---- k_gramb2: a GraphConv layer's BatchNorm backward + both weight-gradient blocks in one pass (sgf_gram2_bn_bwd) ----------
@pytest.mark.parametrize("n", [16384, 16384 + 1, 50001, 262144 + 33])
@pytest.mark.parametrize("m,k", [(256, 256), (128, 128), (64, 64), (256, 104)])
@pytest.mark.parametrize("relu,training", [(True, True), (False, True), (True, False)])
def test_gramb2_matches_the_two_kernels_it_replaces(cuda, n, m, k, relu, training):
    """large/ours.py:36-40,87-93 differentiated: dz = BatchNorm'(relu'(g)) — bit-equal (or one bf16 ulp on a vanishing
    fraction: the two kernels contract their multiply-adds alike, not identically) to sgf_bn_bwd_apply's —, dW blocks
    dz^T y, dz^T x0 and db = sum dz against fp64 of the kernel's own rounded dz (fp32 sums of exact products), and against
    sgf_bn_bwd_apply + sgf_gram2 on the same inputs; identical run to run."""
    from sgformer_amd import ops
    if n > 50001 and (m, k) != (256, 256):
        pytest.skip("large n: the production shape only")
    K = ops.K
    _bn2(True)
    g_ = torch.Generator().manual_seed(n + m + k)
    g = torch.randn(n, m, generator=g_).bfloat16().to(cuda)
    z = (torch.randn(n, m, generator=g_) * 1.3 + 0.2).bfloat16().to(cuda)
    y = torch.randn(n, k, generator=g_).bfloat16().to(cuda)
    x0 = (torch.randn(n, k, generator=g_) + 0.3).bfloat16().to(cuda)
    mean = (torch.randn(m, generator=g_) * 0.2 + 0.2).to(cuda)
    rstd = (1.0 / (1.0 + torch.rand(m, generator=g_))).to(cuda)
    gamma = (1.0 + 0.3 * torch.randn(m, generator=g_)).to(cuda)
    beta = (0.2 * torch.randn(m, generator=g_)).to(cuda)
    stats = K.bn_bwd_stats(g, z, mean, rstd, gamma, beta, relu)
    inv_n = 1.0 / n
    assert K.gram2_bn_bwd_supported(g, z, y, x0)
    dw = torch.empty(m, 2 * k, device=cuda)
    dz, db = K.gram2_bn_bwd(g, z, mean, rstd, gamma, beta, relu, stats, inv_n, training, y, x0, dw[:, :k], dw[:, k:])
    dw2 = torch.empty_like(dw)
    dz2, db2 = K.gram2_bn_bwd(g, z, mean, rstd, gamma, beta, relu, stats, inv_n, training, y, x0, dw2[:, :k], dw2[:, k:])
    assert torch.equal(dz, dz2) and torch.equal(dw, dw2) and torch.equal(db, db2)
    dz_ref = K.bn_bwd_apply(g, z, mean, rstd, gamma, beta, relu, stats, inv_n, training)
    diff = (dz.float() - dz_ref.float()).abs()
    ulp = dz_ref.float().abs() * 2.0 ** -7 + 1e-30
    # (beyond one ulp only where relu's mask sits exactly at its threshold and the two kernels' contractions disagree)
    assert float((diff > ulp).float().mean()) <= 2e-5 and float((diff > 0).float().mean()) <= 2e-3
    dzd = dz.double()
    assert _rel(dw[:, :k], dzd.t() @ y.double()) <= 5e-6
    assert _rel(dw[:, k:], dzd.t() @ x0.double()) <= 5e-6
    assert _rel(db, dzd.sum(0)) <= 5e-6 or float(dzd.sum(0).abs().max()) < 1e-2
    dw3 = torch.empty_like(dw)
    K.gram2(dz_ref, y, x0, dw3[:, :k], dw3[:, k:], want_colsum=False)
    assert _rel(dw, dw3) <= 2e-4


def test_gramb2_strided_operands_refuses_misaligned(cuda):
    from sgformer_amd import ops
    K = ops.K
    _bn2(True)
    n, d = 20000, 256
    g_ = torch.Generator().manual_seed(1)
    wide = torch.randn(n, 4 * d, generator=g_).bfloat16().to(cuda)
    g, z, y, x0 = (wide[:, i * d:(i + 1) * d] for i in range(4))            # column slices: ld = 4 d
    mean, rstd = torch.zeros(d, device=cuda), torch.ones(d, device=cuda)
    stats = K.bn_bwd_stats(g.contiguous(), z.contiguous(), mean, rstd, None, None, True)
    dw = torch.empty(d, 2 * d, device=cuda)
    dz, db = K.gram2_bn_bwd(g, z, mean, rstd, None, None, True, stats, 1.0 / n, True, y, x0, dw[:, :d], dw[:, d:])
    dz_ref = K.bn_bwd_apply(g.contiguous(), z.contiguous(), mean, rstd, None, None, True, stats, 1.0 / n, True)
    assert float((dz.float() - dz_ref.float()).abs().max()) <= 2.0 ** -7 * float(dz_ref.float().abs().max())
    assert _rel(dw[:, :d], dz.double().t() @ y.double()) <= 5e-6
    assert not K.gram2_bn_bwd_supported(g[:, 4:], z[:, 4:], y[:, 4:], x0[:, 4:])     # 8-byte aligned rows only: the old pair


@pytest.mark.parametrize("n,c", [(1, 47), (31, 7), (5000, 47), (70001, 40), (4097, 64), (3000, 2)])
@pytest.mark.parametrize("mapped", [False, True])
def test_head_backward_also_leaves_the_bf16_gradient_operand(cuda, n, c, mapped):
    """sgf_combine_fc_bwd_g (large/ours.py:275 differentiated): dx1 / dx2 bit-equal to sgf_combine_fc_bwd[_mapped]'s, and
    g_out == the logits' gradient cast to bf16, zero-padded to 16 ceil(c / 16) columns, in the module's row order — what the
    cast + pad passes (and the row gather of a re-ordered graph) produced before."""
    from sgformer_amd import ops
    K = ops.K
    d = 256
    g_ = torch.Generator().manual_seed(n + c)
    g = torch.randn(n, c, generator=g_).to(cuda)
    w = (torch.randn(c, d, generator=g_) * 0.1).to(cuda)
    rmap = torch.randperm(n, generator=g_).int().to(cuda) if mapped else None
    dx1, dx2, gp = K.combine_fc_bwd_g(g, w, 0.5, 0.5, rmap)
    r1, r2 = K.combine_fc_bwd(g, w, 0.5, 0.5, torch.bfloat16, rmap) if mapped else K.combine_fc_bwd(g, w, 0.5, 0.5, torch.bfloat16)
    assert torch.equal(dx1, r1) and torch.equal(dx2, r2)
    src = g[rmap.long()] if mapped else g
    cp = (c + 15) // 16 * 16
    assert gp.shape == (n, cp) and gp.dtype == torch.bfloat16
    assert torch.equal(gp[:, :c], src.bfloat16()) and int(torch.count_nonzero(gp[:, c:])) == 0
