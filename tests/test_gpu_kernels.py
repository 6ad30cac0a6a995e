"""Kernel-level parity: every libsgf entry point against the fp64 oracle (oracle/sgformer_oracle.py).

Integer work (CSR build) is checked bit-exact; floating point against fp64 with the tolerance
written at each check.  SURVEY.md §0.5: the attention all-pair term is ~1/(N sqrt(d)) of N*V, so
the attention tests look at the raw partials and at runs with a small `n_total`, where a kernel
that returned V would fail by orders of magnitude.
"""
import numpy as np
import pytest
import torch

from oracle import sgformer_oracle as O

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-300))


# ------------------------------------------------------------------------------------------------
# T1 CSR build: bit-exact
# ------------------------------------------------------------------------------------------------
def _graphs():
    g = torch.Generator().manual_seed(7)
    out = {}
    out["sym_random"] = (O.synthetic_graph(500, 8.0, seed=3), 500)
    out["directed"] = (O.synthetic_graph(300, 5.0, seed=4, directed=True), 300)
    # raw multigraph: duplicates, self loops, isolated nodes, sources with zero in-degree
    src = torch.randint(0, 200, (3000,), generator=g)
    dst = torch.randint(50, 150, (3000,), generator=g)          # nodes <50 and >=150 have in-degree 0
    out["multigraph"] = (torch.stack([src, dst]), 230)            # nodes 200..229 isolated
    out["single_edge"] = (torch.tensor([[1], [0]]), 3)
    out["empty"] = (torch.zeros((2, 0), dtype=torch.int64), 5)
    out["cora_shaped"] = (O.synthetic_graph(2708, 3.9, seed=5), 2708)
    return out


@pytest.mark.parametrize("name", list(_graphs().keys()))
def test_csr_build_bit_exact(cuda, name):
    from sgformer_amd import ops
    ei, n = _graphs()[name]
    rowptr, colind, val, deg = O.csr_build(ei.numpy(), n)
    g = ops.CSRGraph(ei.to(cuda), n)
    assert np.array_equal(g.rowptr.cpu().numpy(), rowptr)
    assert np.array_equal(g.colind.cpu().numpy().astype(np.int64), colind)
    assert np.array_equal(g.deg.cpu().numpy().astype(np.int64), deg)
    # fp32 values: bit pattern equality (IEEE div, sqrt, mul; non-finite -> 0)
    assert np.array_equal(g.val.cpu().numpy().view(np.uint32), val.view(np.uint32))
    t_rowptr, t_colind, t_val, sym = O.csr_transpose(ei.numpy(), n)
    rp, ci, va = g.transposed()
    assert g.symmetric == sym
    assert np.array_equal(rp.cpu().numpy(), t_rowptr)
    assert np.array_equal(ci.cpu().numpy().astype(np.int64), t_colind)
    assert np.array_equal(va.cpu().numpy().view(np.uint32), t_val.view(np.uint32))


def test_csr_rejects_bad_ids(cuda):
    from sgformer_amd import ops
    with pytest.raises(IndexError):
        ops.CSRGraph(torch.tensor([[0, 5], [1, 2]], device=cuda), 4)


# ------------------------------------------------------------------------------------------------
# T2 SpMM
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("d", [4, 32, 64, 100, 128, 256, 512])
@pytest.mark.parametrize("gname", ["sym_random", "multigraph"])
def test_spmm_fp32(cuda, d, gname):
    from sgformer_amd import ops
    ei, n = _graphs()[gname]
    torch.manual_seed(d)
    x = torch.randn(n, d)
    rowptr, colind, val, _ = O.csr_build(ei.numpy(), n)
    ref = O.spmm(rowptr, colind, val, x.double())
    g = ops.CSRGraph(ei.to(cuda), n)
    y = ops.spmm(g, x.to(cuda))
    assert _rel(y, ref) <= 1e-6      # fp32 accumulate vs fp64: ~1e-7 observed
    # backward = A^T dY
    xg = x.to(cuda).requires_grad_(True)
    w = torch.randn(n, d)
    (ops.spmm(g, xg) * w.to(cuda)).sum().backward()
    t_rowptr, t_colind, t_val, _ = O.csr_transpose(ei.numpy(), n)
    gref = O.spmm(t_rowptr, t_colind, t_val, w.double())
    assert _rel(xg.grad, gref) <= 1e-6


def test_spmm_bf16(cuda):
    from sgformer_amd import ops
    ei, n = _graphs()["sym_random"]
    x = torch.randn(n, 256).bfloat16()
    rowptr, colind, val, _ = O.csr_build(ei.numpy(), n)
    ref = O.spmm(rowptr, colind, val, x.double())
    y = ops.spmm(ops.CSRGraph(ei.to(cuda), n), x.to(cuda))
    assert y.dtype == torch.bfloat16
    assert _rel(y.float(), ref) <= 4e-3  # one bf16 rounding of the fp32 accumulator (2^-9)


def test_spmm_into_column_slice(cuda):
    """sgf_spmm with ldx / ldy wider than d: the chunked, pipelined all-gather path of a node-sharded
    run multiplies column chunks of the gathered operand into column slices of ONE output buffer
    (ops._sharded_spmm)."""
    from sgformer_amd import ops
    n, d = 3000, 256
    ei = O.synthetic_graph(n, 9.0, seed=2)
    g = ops.CSRGraph(ei.to(cuda), n)
    x = torch.randn(n, d)
    ref = ops.K.spmm(g.rowptr, g.colind, g.val, x.to(cuda), n)
    for dtype in (torch.float32, torch.bfloat16):
        xg = x.to(cuda, dtype)
        full = ops.K.spmm(g.rowptr, g.colind, g.val, xg, n)
        y = torch.full((n, d), 7.0, dtype=dtype, device=cuda)
        for c in range(4):
            ops.K.spmm(g.rowptr, g.colind, g.val, xg[:, 64 * c:64 * (c + 1)].contiguous(), n,
                       out=y[:, 64 * c:64 * (c + 1)])
        if dtype == torch.float32:
            assert torch.equal(y, full)                  # same per-feature accumulation order
        else:   # bf16 d = 256 pairs even / odd stream entries (k_spmm_seg_bf16x2): another fp32 summation order,
            assert _rel(y.float(), full.float()) <= 2e-3     # so the rounded bf16 results differ by at most an ulp
        assert _rel(y.float(), ref) <= (1e-6 if dtype == torch.float32 else 4e-3)
    with pytest.raises(ValueError):
        ops.K.spmm(g.rowptr, g.colind, g.val, x.to(cuda), n, out=torch.empty(n, d + 4, device=cuda))


@pytest.mark.parametrize("d", [256, 128, 64, 100, 512])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_spmm_long_rows(cuda, d, dtype):
    """Power-law graph: hub rows with thousands of entries go through the split path (sgf_spmm_split:
    queue -> per-segment workgroup partials -> fixed-order combine); every other row through the row
    kernels.  Against the fp64 oracle, and bitwise reproducible from run to run (no float atomics)."""
    from sgformer_amd import ops, synth
    n = 30000
    ei = synth.synthetic_graph_skewed(n, 24.0, gamma=3.0, seed=5)
    rowptr, colind, val, _ = O.csr_build(ei.numpy(), n)
    lens = np.diff(rowptr)
    assert lens.max() > 4 * ops.LONG_ROW          # several segments for the biggest hub
    g = ops.CSRGraph(ei.to(cuda), n)
    exact = int(np.ceil(lens[lens > ops.LONG_ROW] / 1024).sum())
    assert exact > 0 and g.long_segments >= exact   # per-batch-sized graphs: a bound from nnz, no device read-back
    assert ops.long_row_segments(g.rowptr) == exact  # the exact count (graphs above ops.SMALL_GRAPH_NNZ)
    x = torch.randn(n, d, generator=torch.Generator().manual_seed(1)).to(dtype)
    ref = O.spmm(rowptr, colind, val, x.double())
    xg = x.to(cuda)
    y = ops.K.spmm(g.rowptr, g.colind, g.val, xg, n, long_segments=g.long_segments)
    tol = 2e-6 if dtype == torch.float32 else 4e-3
    assert _rel(y.float(), ref) <= tol
    hub = int(lens.argmax())
    assert _rel(y[hub].float(), ref[hub]) <= tol
    y2 = ops.K.spmm(g.rowptr, g.colind, g.val, xg, n, long_segments=g.long_segments)
    assert torch.equal(y, y2)
    # the unsplit kernel gives the same numbers up to fp32 summation order
    y0 = ops.K.spmm(g.rowptr, g.colind, g.val, xg, n, long_segments=0)
    assert _rel(y0.float(), y.float()) <= tol
    # autograd path (symmetric graph: backward reuses the CSR and its long-row plan)
    xr = xg.clone().requires_grad_(True)
    out = ops.spmm(g, xr)
    out.float().sum().backward()
    colsum = torch.zeros(n, dtype=torch.float64).index_add_(0, torch.from_numpy(colind.astype(np.int64)),
                                                           torch.from_numpy(val.astype(np.float64)))
    assert _rel(xr.grad.float()[:, 0], colsum) <= tol


def test_spmm_empty_rows_and_n0(cuda):
    from sgformer_amd import ops
    ei, n = _graphs()["empty"]
    y = ops.spmm(ops.CSRGraph(ei.to(cuda), n), torch.randn(n, 64, device=cuda))
    assert torch.count_nonzero(y) == 0
    g0 = ops.CSRGraph(torch.zeros((2, 0), dtype=torch.int64, device=cuda), 0)
    assert ops.spmm(g0, torch.zeros(0, 64, device=cuda)).shape == (0, 64)


# ------------------------------------------------------------------------------------------------
# T3 attention
# ------------------------------------------------------------------------------------------------
ATTN_CASES = [  # (N, H, d, shared_v)
    (50, 1, 64, False), (777, 1, 64, False), (2708, 1, 64, False), (1000, 1, 128, False),
    (3000, 1, 256, False), (4099, 1, 256, False), (333, 2, 64, False), (333, 2, 64, True),
    (515, 1, 100, False), (129, 3, 32, True), (20000, 1, 256, False),
]


def _qkv(n, h, d, shared_v, seed=0):
    g = torch.Generator().manual_seed(seed + n + 7 * d)
    q = torch.randn(n, h, d, generator=g) * 0.7 + 0.1
    k = torch.randn(n, h, d, generator=g) * 1.3 - 0.2
    v = torch.randn(n, 1 if shared_v else h, d, generator=g)
    return q, k, v


@pytest.mark.parametrize("n,h,d,shared_v", ATTN_CASES)
def test_attention_raw_stats(cuda, n, h, d, shared_v):
    from sgformer_amd import ops
    q, k, v = _qkv(n, h, d, shared_v)
    ref = O.attention_raw_stats(q.double(), k.double(), v.double())
    got = ops.attention_stats(q.to(cuda), k.to(cuda), v.to(cuda))
    nm = h * d * d
    assert _rel(got[:nm], ref[:nm]) <= 2e-6          # S0 = K^T V   (fp32 MFMA chain, two-stage sum)
    assert _rel(got[nm:nm + h * d], ref[nm:nm + h * d]) <= 2e-6   # z0
    assert abs(float(got[-2]) / float(ref[-2]) - 1) <= 2e-6       # ||Q||^2
    assert abs(float(got[-1]) / float(ref[-1]) - 1) <= 2e-6       # ||K||^2


def _run_attention(cuda, q, k, v, n_total, need_grad=True):
    from sgformer_amd import ops
    n, h, d = q.shape
    shared_v = v.shape[1] == 1 and h > 1 or (v.shape[1] == 1 and h == 1 and False)
    qk = torch.cat([q.reshape(n, -1), k.reshape(n, -1)], 1)
    if v.shape[1] == h:
        qkv = torch.cat([qk, v.reshape(n, -1)], 1).to(cuda).requires_grad_(need_grad)
        vx = None
    else:
        qkv = qk.to(cuda).requires_grad_(need_grad)
        vx = v.reshape(n, d).to(cuda).requires_grad_(need_grad)
    out = ops.attention(qkv, vx, h, d, None, n_total)
    return out, qkv, vx


@pytest.mark.parametrize("n,h,d,shared_v", ATTN_CASES)
@pytest.mark.parametrize("small_n", [False, True])
def test_attention_forward_backward(cuda, n, h, d, shared_v, small_n):
    q, k, v = _qkv(n, h, d, shared_v, seed=1)
    n_total = 4.0 if small_n else None       # small n_total: the Q (K^T V) term is O(1) next to N*V
    # (den = qn.z + n_total stays in ~[3, 5]: well conditioned, unlike n_total -> 0)
    qd, kd, vd = (t.double().requires_grad_(True) for t in (q, k, v))
    ref = O.attention(qd, kd, vd, n_total=n_total)
    w = torch.randn(n, d, generator=torch.Generator().manual_seed(5)).double()
    (ref * w).sum().backward()

    out, qkv, vx = _run_attention(cuda, q, k, v, n_total)
    # forward: fp32 vs fp64.  abs tolerance scaled by max|out| (1e-5 relative to the output scale)
    scale = float(ref.detach().abs().max())
    assert float((out.double().cpu() - ref.detach()).abs().max()) <= 1e-5 * scale
    if small_n:
        assert _rel(out, ref.detach()) <= 1e-5
    (out * w.float().to(cuda)).sum().backward()
    hd = h * d
    gq, gk = qkv.grad[:, :hd].reshape(n, h, d), qkv.grad[:, hd:2 * hd].reshape(n, h, d)
    gv = qkv.grad[:, 2 * hd:].reshape(n, h, d) if vx is None else vx.grad.reshape(n, 1, d)
    # gradients: relative (Frobenius) error per tensor; dQ / dK are ~1/N of dV in magnitude
    # (SURVEY.md App. B), so a relative check is the only meaningful one.
    tol = 2e-4 if not small_n else 5e-5
    assert _rel(gv, vd.grad) <= 1e-5
    assert _rel(gq, qd.grad) <= tol
    assert _rel(gk, kd.grad) <= tol


def test_attention_bf16(cuda):
    n, h, d = 3000, 1, 256
    q, k, v = _qkv(n, h, d, False, seed=2)
    qb, kb, vb = (t.bfloat16() for t in (q, k, v))
    ref = O.attention(qb.double(), kb.double(), vb.double(), n_total=4.0)
    out, _, _ = _run_attention(cuda, qb, kb, vb, 4.0, need_grad=False)
    assert out.dtype == torch.bfloat16
    assert _rel(out.float(), ref) <= 8e-3      # bf16 output rounding (2^-8) on fp32 arithmetic


BF16_CASES = [(50, 1, 64, False), (777, 1, 64, False), (1000, 1, 128, False), (3000, 1, 256, False),
              (4099, 1, 256, False), (333, 2, 64, True), (515, 1, 100, False), (20000, 1, 256, False)]


@pytest.mark.parametrize("n,h,d,shared_v", BF16_CASES)
def test_attention_bf16_raw_stats(cuda, n, h, d, shared_v):
    """bf16 storage runs on the bf16 matrix cores (k_reduce_bf16): products of bf16 values are exact
    in fp32 and accumulation is fp32, so against fp64 on the SAME bf16-rounded inputs the partials
    must agree to fp32 accumulation accuracy — this also pins the transposed-LDS fragment layout
    (asymmetric Q / K / V: a row/column swap cannot pass)."""
    from sgformer_amd import ops
    q, k, v = (t.bfloat16() for t in _qkv(n, h, d, shared_v))
    ref = O.attention_raw_stats(q.double(), k.double(), v.double())
    got = ops.attention_stats(q.to(cuda), k.to(cuda), v.to(cuda))
    nm = h * d * d
    assert _rel(got[:nm], ref[:nm]) <= 5e-6
    assert _rel(got[nm:nm + h * d], ref[nm:nm + h * d]) <= 5e-6
    assert abs(float(got[-2]) / float(ref[-2]) - 1) <= 5e-6
    assert abs(float(got[-1]) / float(ref[-1]) - 1) <= 5e-6


@pytest.mark.parametrize("n,h,d,shared_v", BF16_CASES)
def test_attention_bf16_forward_backward(cuda, n, h, d, shared_v):
    """bf16 activations, small n_total so that the all-pair term is O(1): outputs and gradients are
    rounded to bf16 once (2^-9 relative per element) and the d x d matrices enter the apply MFMAs
    rounded to bf16, so the bound is a few 2^-8 in Frobenius norm — a transposed or mis-tiled
    operand would be off by O(1)."""
    q, k, v = (t.bfloat16() for t in _qkv(n, h, d, shared_v, seed=1))
    qd, kd, vd = (t.double().requires_grad_(True) for t in (q, k, v))
    ref = O.attention(qd, kd, vd, n_total=4.0)
    w = torch.randn(n, d, generator=torch.Generator().manual_seed(5)).bfloat16()
    (ref * w.double()).sum().backward()
    out, qkv, vx = _run_attention(cuda, q, k, v, 4.0)
    assert out.dtype == torch.bfloat16
    assert _rel(out.float(), ref.detach()) <= 6e-3
    (out.float() * w.float().to(cuda)).sum().backward()
    hd = h * d
    gq, gk = qkv.grad[:, :hd].reshape(n, h, d), qkv.grad[:, hd:2 * hd].reshape(n, h, d)
    gv = qkv.grad[:, 2 * hd:].reshape(n, h, d) if vx is None else vx.grad.reshape(n, 1, d)
    assert _rel(gv.float(), vd.grad) <= 1e-2
    assert _rel(gq.float(), qd.grad) <= 1.5e-2
    assert _rel(gk.float(), kd.grad) <= 1.5e-2


# ------------------------------------------------------------------------------------------------
# T3 + T4 fused: attention from the un-projected input (sgf_attn_h_*; H = 1)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("n,d,use_wv,small_n", [(50, 64, True, True), (777, 64, True, False), (1000, 128, True, True),
                                                (3000, 256, True, True), (4099, 256, True, False),
                                                (515, 100, True, True), (2000, 256, False, True),
                                                (20000, 256, True, True)])
def test_attention_from_input(cuda, dtype, n, d, use_wv, small_n):
    """out = attention(h Wq^T + bq, h Wk^T + bk, h Wv^T + bv) without materialising Q / K / V, against
    the fp64 oracle applied to explicitly projected Q / K / V; gradients w.r.t. the input AND every
    projection parameter (relative: dWq / dWk are ~1/N of dWv).  small n_total makes
    the all-pair term O(1) so a kernel that ignored it would fail by orders of magnitude."""
    from sgformer_amd import ops
    g = torch.Generator().manual_seed(n + d)
    h = (torch.relu(torch.randn(n, d, generator=g)) * 0.8 + 0.05).to(dtype)     # what the layer sees: >= 0
    ws = [torch.randn(d, d, generator=g) / d ** 0.5 for _ in range(3)]
    bs = [torch.randn(d, generator=g) * 0.1 for _ in range(3)]
    n_total = 4.0 if small_n else None
    wgt = torch.randn(n, d, generator=g).to(dtype)

    hd = h.double().requires_grad_(True)
    wd = [w.double().requires_grad_(True) for w in ws]
    bd = [b.double().requires_grad_(True) for b in bs]
    q = (hd @ wd[0].t() + bd[0]).unsqueeze(1)
    k = (hd @ wd[1].t() + bd[1]).unsqueeze(1)
    v = (hd @ wd[2].t() + bd[2]).unsqueeze(1) if use_wv else hd.unsqueeze(1)
    ref = O.attention(q, k, v, n_total=n_total)
    (ref * wgt.double()).sum().backward()

    hg = h.to(cuda).requires_grad_(True)
    wg = [w.to(cuda).requires_grad_(True) for w in ws]
    bg = [b.to(cuda).requires_grad_(True) for b in bs]
    out = ops.attention_from_input(hg, wg[0], bg[0], wg[1], bg[1], wg[2] if use_wv else None,
                                   bg[2] if use_wv else None, None, n_total)
    assert out.dtype == dtype
    (out.float() * wgt.float().to(cuda)).sum().backward()
    f32 = dtype == torch.float32
    assert _rel(out.float(), ref.detach()) <= (2e-6 if f32 else 6e-3)
    assert _rel(hg.grad.float(), hd.grad) <= (2e-4 if f32 else 1.5e-2)
    names = ["q", "k", "v"] if use_wv else ["q", "k"]
    for i, nm in enumerate(names):
        tol = (2e-4 if f32 else 2e-2)
        assert wg[i].grad.dtype == torch.float32
        assert _rel(wg[i].grad, wd[i].grad) <= tol, nm
        assert _rel(bg[i].grad, bd[i].grad) <= tol, nm
    if not use_wv:
        assert wg[2].grad is None and bg[2].grad is None


# ------------------------------------------------------------------------------------------------
# T4 weight / bias gradients: sgf_gram
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("n,m,k", [(5000, 256, 256), (3000, 768, 256), (1000, 256, 100), (777, 64, 64),
                                   (50, 128, 36), (2049, 256, 512), (300, 4, 8), (0, 64, 32)])
def test_gram(cuda, dtype, n, m, k):
    from sgformer_amd import ops
    g = torch.Generator().manual_seed(n + m + k)
    a = (torch.randn(n, m, generator=g) * 0.5 + 0.3).to(dtype)
    b = (torch.randn(n, k, generator=g) * 1.5 - 0.2).to(dtype)
    c, cs = ops.K.gram(a.to(cuda), b.to(cuda))
    ref = a.double().t() @ b.double()
    assert c.dtype == torch.float32 and c.shape == (m, k)
    if n == 0:
        assert torch.count_nonzero(c) == 0 and torch.count_nonzero(cs) == 0
        return
    assert _rel(c, ref) <= 5e-6                      # fp32 accumulation of exact (bf16) / fp32 products
    assert _rel(cs, a.double().sum(0)) <= 5e-6
    # column-sliced output view (how ops.linear_cat fills dW = [dW_1 | dW_2])
    big = torch.full((m, k + 12), 7.0, device=cuda)
    ops.K.gram(a.to(cuda), b.to(cuda), out=big[:, 4:4 + k], want_colsum=False)
    assert _rel(big[:, 4:4 + k], ref) <= 5e-6
    assert bool((big[:, :4] == 7.0).all()) and bool((big[:, 4 + k:] == 7.0).all())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("widths,out_dim", [((256,), 256), ((256, 256), 256), ((100,), 256), ((64,), 47)])
def test_linear_gradients(cuda, dtype, widths, out_dim):
    """ops.linear / linear_cat (hipBLASLt forward + dX, sgf_gram dW / db) against autograd of the
    plain fp64 expression; fp32 master weights, activations in `dtype`."""
    from sgformer_amd import ops
    g = torch.Generator().manual_seed(11)
    n = 1500
    xs = [torch.randn(n, w, generator=g).to(dtype) for w in widths]
    w = torch.randn(out_dim, sum(widths), generator=g) * 0.1
    b = torch.randn(out_dim, generator=g)
    gy = torch.randn(n, out_dim, generator=g).to(dtype)
    xd = [x.double().requires_grad_(True) for x in xs]
    wd, bd = w.double().requires_grad_(True), b.double().requires_grad_(True)
    ref = torch.cat(xd, 1) @ wd.t() + bd
    ref.backward(gy.double())
    xg = [x.to(cuda).requires_grad_(True) for x in xs]
    wg, bg = w.to(cuda).requires_grad_(True), b.to(cuda).requires_grad_(True)
    y = ops.linear_cat(tuple(xg), wg, bg)
    y.backward(gy.to(cuda))
    tol = 1e-5 if dtype == torch.float32 else 8e-3
    assert _rel(y.float(), ref.detach()) <= tol
    assert wg.grad.dtype == torch.float32 and bg.grad.dtype == torch.float32
    assert _rel(wg.grad, wd.grad) <= (1e-5 if dtype == torch.float32 else 1e-5)  # exact products, fp32 acc
    assert _rel(bg.grad, bd.grad) <= 1e-5
    for a, r in zip(xg, xd):
        assert _rel(a.grad.float(), r.grad) <= tol


def test_attention_n0(cuda):
    from sgformer_amd import ops
    out = ops.attention(torch.zeros(0, 3 * 64, device=cuda), None, 1, 64)
    assert out.shape == (0, 64)


# ------------------------------------------------------------------------------------------------
# T5 / T6 / T7 fused glue
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("d", [64, 100, 256, 512])
@pytest.mark.parametrize("relu,use_ln,use_res", [(True, True, True), (False, True, True),
                                                  (True, True, False), (True, False, True)])
def test_ln_res_act(cuda, d, relu, use_ln, use_res):
    from sgformer_amd import ops
    torch.manual_seed(d)
    n = 777
    x, r = torch.randn(n, d) * 2 + 0.3, torch.randn(n, d)
    gamma, beta = 1 + 0.1 * torch.randn(d), 0.1 * torch.randn(d)
    a, b = 0.3, 0.7
    w = torch.randn(n, d)

    def ref_fn(x, r, gamma, beta):
        pre = a * x + (b * r if use_res else 0)
        if use_ln:
            pre = torch.nn.functional.layer_norm(pre, (d,), gamma, beta, 1e-5)
        return torch.relu(pre) if relu else pre

    xs = [t.double().requires_grad_(True) for t in (x, r, gamma, beta)]
    (ref_fn(*xs) * w.double()).sum().backward()
    gs = [t.to(cuda).requires_grad_(True) for t in (x, r, gamma, beta)]
    y = ops.ln_res_act(gs[0], gs[1] if use_res else None, a, b, gs[2] if use_ln else None,
                       gs[3] if use_ln else None, relu, 1e-5)
    assert float((y.double().cpu() - ref_fn(*[t.detach() for t in xs])).abs().max()) <= 2e-5
    (y * w.to(cuda)).sum().backward()
    assert _rel(gs[0].grad, xs[0].grad) <= 1e-5
    if use_res:
        assert _rel(gs[1].grad, xs[1].grad) <= 1e-5
    if use_ln:
        assert _rel(gs[2].grad, xs[2].grad) <= 1e-5
        assert _rel(gs[3].grad, xs[3].grad) <= 1e-5


@pytest.mark.parametrize("d", [64, 100, 256])
@pytest.mark.parametrize("relu,use_res,training", [(True, True, True), (False, False, True),
                                                   (True, True, False)])
def test_bn_act_res(cuda, d, relu, use_res, training):
    from sgformer_amd import ops
    torch.manual_seed(d + 1)
    n = 1531
    x, r = torch.randn(n, d) * 1.7 + 5.0, torch.randn(n, d)     # large mean: two-pass variance
    gamma, beta = 1 + 0.1 * torch.randn(d), 0.1 * torch.randn(d)
    rm, rv = 5 + 0.1 * torch.randn(d), 2.5 + 0.2 * torch.rand(d)
    w = torch.randn(n, d)

    def ref_fn(x, r, gamma, beta):
        y = torch.nn.functional.batch_norm(x, rm.double(), rv.double(), gamma, beta, training, 0.0, 1e-5)
        if relu:
            y = torch.relu(y)
        return y + r if use_res else y

    xs = [t.double().requires_grad_(True) for t in (x, r, gamma, beta)]
    (ref_fn(*xs) * w.double()).sum().backward()
    gs = [t.to(cuda).requires_grad_(True) for t in (x, r, gamma, beta)]
    if training:
        mean, var, n_tot = ops.batch_stats(gs[0].detach())
        assert _rel(mean, x.double().mean(0)) <= 1e-6
        assert _rel(var, x.double().var(0, unbiased=False)) <= 1e-5
    else:
        mean, var, n_tot = rm.to(cuda), rv.to(cuda), float(n)
    rstd = torch.rsqrt(var + 1e-5)
    y = ops.bn_act_res(gs[0], gs[1] if use_res else None, gs[2], gs[3], mean, rstd, relu, training, n_tot)
    assert float((y.double().cpu() - ref_fn(*[t.detach() for t in xs])).abs().max()) <= 2e-5
    (y * w.to(cuda)).sum().backward()
    assert _rel(gs[0].grad, xs[0].grad) <= 2e-5
    assert _rel(gs[2].grad, xs[2].grad) <= 1e-5
    assert _rel(gs[3].grad, xs[3].grad) <= 1e-5
    if use_res:
        assert _rel(gs[1].grad, xs[1].grad) <= 1e-6


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_axpby(cuda, dtype):
    """large/ours.py:269-270 `graph_weight * x2 + (1 - graph_weight) * x1`: forward and both gradients
    against the same arithmetic in fp64 on the HOST (never against torch on the GPU)."""
    from sgformer_amd import ops
    g = torch.Generator().manual_seed(5)
    a, b = torch.randn(1000, 256, generator=g).to(dtype), torch.randn(1000, 256, generator=g).to(dtype)
    w = torch.randn(1000, 256, generator=g)
    ag, bg = a.to(cuda).requires_grad_(True), b.to(cuda).requires_grad_(True)
    y = ops.axpby(ag, bg, 0.8, 0.2)
    ref = 0.8 * a.double() + 0.2 * b.double()
    tol = 1e-6 if dtype == torch.float32 else 2.0 ** -8
    assert float((y.double().cpu() - ref).abs().max()) <= tol * max(1.0, float(ref.abs().max()))
    (y.float() * w.to(cuda)).sum().backward()
    assert _rel(ag.grad.float(), 0.8 * w.double()) <= (1e-6 if dtype == torch.float32 else 4e-3)
    assert _rel(bg.grad.float(), 0.2 * w.double()) <= (1e-6 if dtype == torch.float32 else 4e-3)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("k", [2, 3, 7, 8])
def test_sum_n_and_fan_out(cuda, dtype, k):
    from sgformer_amd import ops
    g = torch.Generator().manual_seed(k)
    xs = [torch.randn(999, 100, generator=g).to(dtype) for _ in range(k)]
    got = ops.K.sum_n([x.to(cuda) for x in xs])
    ref = sum(x.double() for x in xs)
    assert got.dtype == dtype
    assert _rel(got.float(), ref) <= (1e-6 if dtype == torch.float32 else 4e-3)
    # the hub: k consumers, one fused gradient sum
    x = xs[0].to(cuda).requires_grad_(True)
    outs = ops.fan_out(x, k)
    ws = [torch.randn(999, 100, generator=g).to(dtype).to(cuda) for _ in range(k)]
    sum((o.float() * w.float()).sum() for o, w in zip(outs, ws)).backward()
    assert _rel(x.grad.float(), sum(w.double().cpu() for w in ws)) <= (1e-6 if dtype == torch.float32 else 4e-3)


# ------------------------------------------------------------------------------------------------
# N1 induced subgraph (sgf_subgraph_*): bit-exact against the PyG 1.7.2 semantics
# ------------------------------------------------------------------------------------------------
def _ref_subgraph(subset, ei, n, relabel):
    mask_n = torch.zeros(n, dtype=torch.bool)
    mask_n[subset] = True
    keep = mask_n[ei[0]] & mask_n[ei[1]]
    out = ei[:, keep]
    if relabel:
        idx = torch.zeros(n, dtype=torch.int64)
        idx[subset] = torch.arange(subset.numel())
        out = idx[out]
    return out, keep


@pytest.mark.parametrize("n,nnz,m", [(1000, 20000, 300), (50000, 1500000, 7000), (257, 513, 257), (100, 4000, 0),
                                     (3000, 0, 50), (200000, 3000001, 60000)])
@pytest.mark.parametrize("relabel", [True, False])
def test_subgraph_bit_exact(cuda, n, nnz, m, relabel):
    from sgformer_amd import batching
    g = torch.Generator().manual_seed(n + nnz + m)
    ei = torch.randint(0, n, (2, nnz), generator=g)            # duplicates and self-loops included
    subset = torch.randperm(n, generator=g)[:m]
    ref, keep = _ref_subgraph(subset, ei, n, relabel)
    w = torch.randn(nnz, generator=g)
    # host edge_index (what the trainer passes): staged once, result on the GPU
    out, attr = batching.subgraph(subset, ei, edge_attr=w, relabel_nodes=relabel, num_nodes=n)
    assert out.is_cuda and out.dtype == torch.int64 and out.shape == ref.shape
    assert torch.equal(out.cpu(), ref)
    assert torch.equal(attr.cpu(), w[keep])
    # device inputs, bool-mask subset, no attributes
    mask = torch.zeros(n, dtype=torch.bool)
    mask[subset] = True
    out2, none = batching.subgraph(mask.to(cuda), ei.to(cuda), relabel_nodes=False, num_nodes=n)
    assert none is None and torch.equal(out2.cpu(), _ref_subgraph(subset, ei, n, False)[0])


def test_subgraph_feeds_the_model(cuda):
    """large/main-batch.py:138-143 on the GPU: gather the batch rows, cut the induced subgraph, run
    the model on it — identical logits to the host-side PyG-semantics cut."""
    from sgformer_amd import batching
    from sgformer_amd.ours import SGFormer
    n, f, d, c = 4000, 16, 64, 5
    torch.manual_seed(0)
    x = torch.randn(n, f)
    ei = O.synthetic_graph(n, 12.0, seed=3)
    idx = torch.randperm(n)[:1500]
    m = SGFormer(f, d, c, trans_dropout=0.0, gnn_dropout=0.0, gnn_num_layers=2, gnn_use_init=True).to(cuda).eval()
    ei_ref, _ = _ref_subgraph(idx, ei, n, True)
    ei_gpu, _ = batching.subgraph(idx, ei, num_nodes=n, relabel_nodes=True)
    with torch.no_grad():
        a = m(x[idx].to(cuda), ei_ref.to(cuda))
        b = m(x[idx].to(cuda), ei_gpu)
    assert torch.equal(a, b)


# ------------------------------------------------------------------------------------------------
# N4 fused log_softmax + NLL on the training rows (sgf_nll_*)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("n,c,m", [(1000, 2, 400), (5000, 7, 140), (30000, 47, 15000), (4000, 172, 4000),
                                   (700, 300, 350), (50, 40, 0)])
def test_fused_loss(cuda, dtype, n, c, m):
    from sgformer_amd.loss import log_softmax_nll
    g = torch.Generator().manual_seed(n + c)
    logits = (torch.randn(n, c, generator=g) * 3).to(dtype)
    y = torch.randint(0, c, (n, 1), generator=g)               # [N, 1] as the trainers keep it
    idx = torch.randperm(n, generator=g)[:m]
    ld = logits.double().requires_grad_(True)
    if m > 0:
        ref = O.nll_loss(ld, y.squeeze(1), idx)
        ref.backward()
    lg = logits.to(cuda).requires_grad_(True)
    loss = log_softmax_nll(lg, y.to(cuda), idx.to(cuda))
    loss.backward()
    assert loss.dtype == torch.float32 and lg.grad.dtype == dtype and lg.grad.shape == (n, c)
    if m == 0:
        assert float(loss) == 0.0 and torch.count_nonzero(lg.grad) == 0
        return
    assert abs(float(loss) - float(ref)) <= 2e-6 * abs(float(ref)) + 1e-6
    assert _rel(lg.grad.float(), ld.grad) <= (2e-6 if dtype == torch.float32 else 5e-3)
    off = torch.ones(n, dtype=torch.bool)
    off[idx] = False
    assert torch.count_nonzero(lg.grad[off.to(cuda)]) == 0      # exact zeros off the training rows
    # bool-mask index + explicit divisor (node-sharded form)
    mask = ~off
    l2 = log_softmax_nll(lg.detach(), y.to(cuda), mask.to(cuda), denom=2 * m)
    assert abs(float(l2) * 2 - float(ref)) <= 2e-6 * abs(float(ref)) + 1e-6


# ------------------------------------------------------------------------------------------------
# dropout + residual without a stored mask (sgf_dropout)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("p", [0.2, 0.5, 0.9])
def test_fused_dropout(cuda, dtype, p):
    """F.dropout semantics (kept elements scaled by 1/(1-p), the rest zero) + residual; the backward
    recomputes EXACTLY the forward's keep pattern from the seed; keep rate, per-row and per-column
    rates within binomial bounds (a counter bug would show as stripes); reproducible under
    torch.manual_seed, different from seed to seed."""
    from sgformer_amd import ops
    n, d = 5000, 256
    g = torch.Generator().manual_seed(0)
    x = (torch.rand(n, d, generator=g) + 0.5).to(dtype)
    res = (torch.rand(n, d, generator=g) * 2 - 1).to(dtype)
    w = (torch.rand(n, d, generator=g) + 0.5).to(dtype)
    xg, rg = x.to(cuda).requires_grad_(True), res.to(cuda).requires_grad_(True)
    torch.manual_seed(7)
    y = ops.dropout_res(xg, rg, p)
    y.backward(w.to(cuda))
    keep = (y.detach() != rg.detach())
    rate = float(keep.float().mean())
    assert abs(rate - (1 - p)) <= 5 * (p * (1 - p) / (n * d)) ** 0.5
    assert float((keep.float().mean(0) - (1 - p)).abs().max()) <= 6 * (p * (1 - p) / n) ** 0.5
    assert float((keep.float().mean(1) - (1 - p)).abs().max()) <= 6 * (p * (1 - p) / d) ** 0.5
    expect = torch.where(keep, xg.detach().float() / (1 - p), torch.zeros((), device=cuda)) + rg.detach().float()
    assert _rel(y.detach().float(), expect) <= (1e-6 if dtype == torch.float32 else 4e-3)
    gx_expect = torch.where(keep, w.to(cuda).float() / (1 - p), torch.zeros((), device=cuda))
    assert _rel(xg.grad.float(), gx_expect) <= (1e-6 if dtype == torch.float32 else 4e-3)
    assert torch.equal(xg.grad != 0, keep)                       # same pattern, element for element
    assert torch.equal(rg.grad, w.to(cuda))
    torch.manual_seed(7)
    assert torch.equal(ops.dropout_res(xg.detach(), rg.detach(), p), y.detach())
    y2 = ops.dropout_res(xg.detach(), rg.detach(), p)            # next seed of the stream
    assert float(((y2 != rg.detach()) != keep).float().mean()) > 0.5 * min(p, 1 - p)
    y3 = ops.dropout_res(xg.detach(), None, p)
    assert float(((y3 != 0).float().mean())) == pytest.approx(1 - p, abs=0.01)


def test_cpu_tensor_is_rejected():
    from sgformer_amd import ops
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.attention(torch.zeros(4, 192), None, 1, 64)


@pytest.mark.parametrize("d", [1, 7, 47, 172, 256])
def test_colsum_any_width(cuda, d):
    from sgformer_amd import ops
    torch.manual_seed(d)
    x = torch.randn(5003, d)
    got = ops.K.colsum(x.to(cuda))
    assert _rel(got, x.double().sum(0)) <= 1e-6
    assert float(ops.K.colsum(torch.zeros(0, d, device=cuda)).abs().max()) == 0.0


@pytest.mark.parametrize("n,d,c", [(1000, 256, 47), (777, 64, 7), (3001, 128, 40), (4097, 256, 172), (500, 256, 2), (1, 64, 5)])
def test_combine_fc_fused_fp32(cuda, n, d, c):
    """T7 with fp32 storage (BASELINE.json configs 2 and 4), large/ours.py:269-270,275: logits = fc(gw x2 + (1 - gw) x1) as
    ONE kernel on the exact-fp32 matrix cores (csrc/linear_f32.hip: the combination is formed while the row tile is staged,
    never written; class counts that are not multiples of 4 run with zero-padded rows of W), and its backward as one kernel
    that stores both scaled copies of dlogits W.  Against fp64: 2e-6 relative (summation order only)."""
    from sgformer_amd import ops
    g = torch.Generator().manual_seed(n + d + c)
    x1, x2 = torch.randn(n, d, generator=g), torch.randn(n, d, generator=g)
    w = torch.randn(c, d, generator=g) / d ** 0.5
    b = torch.randn(c, generator=g) * 0.1
    go = torch.randn(n, c, generator=g)
    gw = 0.8
    assert ops.combine_fc_supported(x1.to(cuda), c)
    x1g, x2g = x1.to(cuda).requires_grad_(True), x2.to(cuda).requires_grad_(True)
    wg, bg = w.to(cuda).requires_grad_(True), b.to(cuda).requires_grad_(True)
    out = ops.combine_fc(x2g, x1g, wg, bg, gw, 1.0 - gw)
    assert out.dtype == torch.float32 and out.shape == (n, c)
    (out * go.to(cuda)).sum().backward()
    xc = gw * x2.double() + (1.0 - gw) * x1.double()
    ref = xc @ w.double().t() + b.double()
    assert _rel(out, ref) <= 2e-6
    dx = go.double() @ w.double()
    assert _rel(x2g.grad, gw * dx) <= 2e-6 and _rel(x1g.grad, (1.0 - gw) * dx) <= 2e-6
    assert _rel(wg.grad, go.double().t() @ xc) <= 1e-5
    assert _rel(bg.grad, go.double().sum(0)) <= 1e-5 or n < 10


@pytest.mark.parametrize("n,d,c", [(3000, 128, 172), (1025, 256, 172), (500, 256, 65), (2000, 64, 256)])
def test_combine_fc_fused_bf16_many_classes(cuda, monkeypatch, n, d, c):
    """T7 for bf16 activations and MORE than 64 classes (C = 172: the papers100M recipe, 100M/run.sh:3-7 — BASELINE.json
    config 5): the same single kernel as the fp32 head with bf16 rows on the wire — combination and product in exact fp32
    (no rounding of the combined activations at all), fp32 logits; backward stores both scaled copies of dlogits W as bf16.
    Against fp64 of the same bf16 inputs: logits 2e-6, dx one bf16 rounding, dW / db 2e-5."""
    from sgformer_amd import ops
    g = torch.Generator().manual_seed(n + d + c)
    x1, x2 = torch.randn(n, d, generator=g).bfloat16(), torch.randn(n, d, generator=g).bfloat16()
    w = torch.randn(c, d, generator=g) / d ** 0.5
    b = torch.randn(c, generator=g) * 0.1
    go = torch.randn(n, c, generator=g)
    gw = 0.8
    # the module's policy keeps this shape on sgf_axpby + the bf16 Linear (faster: DESIGN §3.5); the one-kernel form is opt-in
    assert not ops.combine_fc_supported(x1.to(cuda), c)
    monkeypatch.setenv("SGF_HEAD_WIDE", "1")
    assert ops.combine_fc_supported(x1.to(cuda), c)
    x1g, x2g = x1.to(cuda).requires_grad_(True), x2.to(cuda).requires_grad_(True)
    wg, bg = w.to(cuda).requires_grad_(True), b.to(cuda).requires_grad_(True)
    out = ops.combine_fc(x2g, x1g, wg, bg, gw, 1.0 - gw)
    assert out.dtype == torch.float32 and out.shape == (n, c)
    (out * go.to(cuda)).sum().backward()
    xc = gw * x2.double() + (1.0 - gw) * x1.double()
    ref = xc @ w.double().t() + b.double()
    assert _rel(out, ref) <= 2e-6
    dx = go.double() @ w.double()
    assert x1g.grad.dtype == torch.bfloat16
    assert _rel(x2g.grad.float(), gw * dx) <= 4e-3 and _rel(x1g.grad.float(), (1.0 - gw) * dx) <= 4e-3
    go_r = go.bfloat16().double()                      # the weight gradient sees dlogits in the activation dtype
    assert _rel(wg.grad, go_r.t() @ xc) <= 2e-5
    assert _rel(bg.grad, go_r.sum(0)) <= 2e-5


@pytest.mark.parametrize("n,d,c", [(1000, 256, 47), (777, 64, 7), (3001, 128, 40), (33, 256, 64), (500, 256, 2)])
def test_combine_fc_fused(cuda, n, d, c):
    """T7, large/ours.py:269-270,275: logits = fc(gw * x2 + (1 - gw) * x1) in one kernel (bf16 activations),
    forward and all four gradients against fp64 ON THE HOST of the same arithmetic applied to the same
    bf16-rounded inputs (combined activations rounded to bf16 once, weights rounded to bf16)."""
    from sgformer_amd import ops
    g = torch.Generator().manual_seed(n + d + c)
    x1 = torch.randn(n, d, generator=g).bfloat16()
    x2 = torch.randn(n, d, generator=g).bfloat16()
    w = (torch.randn(c, d, generator=g) / d ** 0.5)
    b = torch.randn(c, generator=g) * 0.1
    go = torch.randn(n, c, generator=g)
    gw = 0.8
    assert ops.combine_fc_supported(x1.to(cuda), c)
    x1g, x2g = x1.to(cuda).requires_grad_(True), x2.to(cuda).requires_grad_(True)
    wg, bg = w.to(cuda).requires_grad_(True), b.to(cuda).requires_grad_(True)
    out = ops.combine_fc(x2g, x1g, wg, bg, gw, 1.0 - gw)          # the module's argument order
    assert out.dtype == torch.float32 and out.shape == (n, c)
    (out * go.to(cuda)).sum().backward()
    # reference arithmetic on the host: the combination is formed in fp32 (as sgf_axpby forms it) and rounded to
    # bf16 once, everything after that in fp64.  Forming it in fp64 instead flips the bf16 rounding of ~0.25 % of
    # the elements (bf16 inputs times 0.8 / 0.2 land near rounding boundaries), the kernel's fused multiply-add a
    # handful more: so the MEAN error is the criterion (exact rows dominate), the max is bounded by one bf16 ulp
    # of one element times a weight.
    xc = (gw * x2.double() + (1.0 - gw) * x1.double())
    xc_r = (torch.tensor(gw, dtype=torch.float32) * x2.float()
            + torch.tensor(1.0 - gw, dtype=torch.float32) * x1.float()).bfloat16().double()
    w_r = w.bfloat16().double()
    ref = xc_r @ w_r.t() + b.double()
    err = (out.double().cpu() - ref).abs()
    scale = max(1.0, float(ref.abs().max()))
    assert float(err.mean()) <= 1e-4 * scale and float(err.max()) <= 4e-3 * scale, (float(err.mean()), float(err.max()))
    go_r = go.bfloat16().double()                                  # logits gradient rounded to the activation dtype
    dx = go_r @ w_r
    assert _rel(x2g.grad.float(), gw * dx) <= 4e-3 and _rel(x1g.grad.float(), (1.0 - gw) * dx) <= 4e-3
    assert x1g.grad.dtype == torch.bfloat16
    dw = go_r.t() @ xc                                             # fp32-exact products of bf16 values
    assert _rel(wg.grad, dw) <= 2e-5
    assert _rel(bg.grad, go_r.sum(0)) <= 2e-5
    # and the unfused path (sgf_axpby + library GEMM) agrees to bf16 rounding of the logits
    unf = ops.out_linear(ops.axpby(x2.to(cuda), x1.to(cuda), gw, 1.0 - gw), w.to(cuda), b.to(cuda)).float()
    assert float((unf.cpu() - out.detach().cpu()).abs().max()) <= 2.0 ** -7 * max(1.0, float(ref.abs().max()))


# ------------------------------------------------------------------------------------------------
# T6 / K8: Linear (+ BatchNorm statistics) as one streaming pass (csrc/rowgemm.hip)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n,d", [(1, 64), (31, 256), (32, 128), (77, 64), (1000, 256), (4097, 128), (20001, 256)])
def test_gcn_epilogue_stats_and_dx(cuda, n, d):
    """large/ours.py:36-40 (x = self.W(x)) and the batch statistics of :87-88 in one pass, against fp64 ON THE HOST
    of the same bf16-rounded operands: y within one bf16 rounding of the fp64 product, the statistics equal to the
    fp64 column sums of the y the kernel RETURNED (rel 2e-6: fp32 accumulation of at most 20 001 terms), dx = dy W."""
    from sgformer_amd import ops
    g = torch.Generator().manual_seed(13 * n + d)
    a = torch.randn(n, d, generator=g).bfloat16()
    w = (torch.randn(d, d, generator=g) / d ** 0.5).bfloat16()
    bias = torch.randn(d, generator=g)
    shift = torch.randn(d, generator=g) * 0.1
    assert ops.K.gcn_epilogue_supported(d, d, torch.bfloat16)
    y, st = ops.K.gcn_epilogue_stats(a.to(cuda), w.to(cuda), bias.to(cuda), shift.to(cuda), want_stats=True)
    y2, none = ops.K.gcn_epilogue_stats(a.to(cuda), w.to(cuda), bias.to(cuda))
    assert none is None and torch.equal(y, y2) and y.dtype == torch.bfloat16
    y3, st3 = ops.K.gcn_epilogue_stats(a.to(cuda), w.to(cuda), bias.to(cuda), shift.to(cuda), want_stats=True)
    assert torch.equal(y, y3) and torch.equal(st, st3)            # two-stage fixed-order sums: run-to-run identical
    ref = a.double() @ w.double().t() + bias.double()
    err = (y.double().cpu() - ref).abs()
    assert bool((err <= 2.0 ** -8 * ref.abs() + 1e-6).all()), float((err - 2.0 ** -8 * ref.abs()).max())
    v = y.double().cpu() - shift.double()
    st_ref = torch.cat([v.sum(0), (v * v).sum(0)])
    tol = 2e-6 * torch.cat([v.abs().sum(0), (v * v).sum(0)]).clamp_min(1e-3)
    assert bool(((st.double().cpu() - st_ref).abs() <= tol).all())
    # no shift, no bias
    y0, st0 = ops.K.gcn_epilogue_stats(a.to(cuda), w.to(cuda), None, None, want_stats=True)
    v0 = y0.double().cpu()
    assert bool(((y0.double().cpu() - a.double() @ w.double().t()).abs() <= 2.0 ** -8 * v0.abs() + 1e-6).all())
    assert _rel(st0[d:], (v0 * v0).sum(0)) <= 2e-6
    # dx = dy W (contraction over W's rows)
    dx = ops.K.gcn_epilogue_dx(a.to(cuda), w.to(cuda))
    refdx = a.double() @ w.double()
    errdx = (dx.double().cpu() - refdx).abs()
    assert bool((errdx <= 2.0 ** -8 * refdx.abs() + 1e-6).all())


@pytest.mark.parametrize("n,d", [(100, 256), (4096, 256), (4097, 128), (70001, 256), (33000, 64)])
def test_gcn_epilogue_dx2_paired_launch(cuda, n, d):
    """Both input gradients of W [a1 | a2] (large/ours.py:36-38) from one launch whose workgroups come in pairs on one XCD
    (sgf_gcn_epilogue_dx2): bit-identical to the two separate sgf_gcn_epilogue_dx launches, and dy W[:, d:] against fp64
    of the same bf16 operands within one bf16 rounding.  n covers: too few tiles to pair (falls back), whole groups of
    pairs, a ragged last tile."""
    from sgformer_amd import ops
    g = torch.Generator().manual_seed(n + d)
    dy = torch.randn(n, d, generator=g).bfloat16().to(cuda)
    w = (torch.randn(d, 2 * d, generator=g) / d ** 0.5).bfloat16().to(cuda)
    a1, a2 = ops.K.gcn_epilogue_dx2(dy, w[:, :d], w[:, d:], pair=True)
    b1, b2 = ops.K.gcn_epilogue_dx2(dy, w[:, :d], w[:, d:], pair=False)
    assert torch.equal(a1, b1) and torch.equal(a2, b2)
    assert torch.equal(b1, ops.K.gcn_epilogue_dx(dy, w[:, :d]))
    ref = dy.double().cpu() @ w[:, d:].double().cpu()
    err = (a2.double().cpu() - ref).abs()
    assert bool((err <= 2.0 ** -8 * ref.abs() + 1e-6).all())


@pytest.mark.parametrize("n,d_in,d_out", [(3001, 256, 256), (517, 128, 256), (1000, 100, 64), (333, 256, 40), (5, 64, 64),
                                          (2111, 128, 128)])
def test_linear_f32_stats_dx_and_cat(cuda, n, d_in, d_out):
    """fp32 storage (BASELINE.json configs 2 and 4): the Linear layers of large/ours.py:36-40, :77, :198, :275, the batch
    statistics of :87-88 and dX on the exact-fp32 matrix cores (csrc/linear_f32.hip) against fp64: 1e-6 relative (the
    fp32 MFMA is an exact FMA chain — only the summation order differs from a CPU loop), statistics 2e-6 of the column
    sums of the RETURNED y, run-to-run identical."""
    from sgformer_amd import ops
    g = torch.Generator().manual_seed(7 * n + d_in)
    a = torch.randn(n, d_in, generator=g)
    w = torch.randn(d_out, d_in, generator=g) / d_in ** 0.5
    bias, shift = torch.randn(d_out, generator=g), torch.randn(d_out, generator=g) * 0.1
    assert ops.K.gcn_epilogue_supported(d_in, d_out, torch.float32)
    y, st = ops.K.gcn_epilogue_stats(a.to(cuda), w.to(cuda), bias.to(cuda), shift.to(cuda), want_stats=True)
    y2, st2 = ops.K.gcn_epilogue_stats(a.to(cuda), w.to(cuda), bias.to(cuda), shift.to(cuda), want_stats=True)
    assert y.dtype == torch.float32 and torch.equal(y, y2) and torch.equal(st, st2)
    ref = a.double() @ w.double().t() + bias.double()
    assert _rel(y, ref) <= 1e-6
    v = y.double().cpu() - shift.double()
    st_ref = torch.cat([v.sum(0), (v * v).sum(0)])
    tol = 2e-6 * torch.cat([v.abs().sum(0), (v * v).sum(0)]).clamp_min(1e-3)
    assert bool(((st.double().cpu() - st_ref).abs() <= tol).all())
    y0, none = ops.K.gcn_epilogue_stats(a.to(cuda), w.to(cuda), None)
    assert none is None and _rel(y0, a.double() @ w.double().t()) <= 1e-6
    gy = torch.randn(n, d_out, generator=g)
    dx = ops.K.gcn_epilogue_dx(gy.to(cuda), w.to(cuda))
    assert dx.shape == (n, d_in) and _rel(dx, gy.double() @ w.double()) <= 1e-6
    # two-operand form [a | a2] W^T + b with W's column blocks taken as strided slices
    a2 = torch.randn(n, d_out, generator=g)
    wc = torch.randn(d_out, d_in + d_out, generator=g) / (d_in + d_out) ** 0.5
    yc, stc = ops.K.gcn_epilogue_cat(a.to(cuda), a2.to(cuda), wc.to(cuda), bias.to(cuda), shift.to(cuda), want_stats=True)
    refc = torch.cat([a, a2], 1).double() @ wc.double().t() + bias.double()
    assert _rel(yc, refc) <= 1e-6
    vc = yc.double().cpu() - shift.double()
    assert _rel(stc[:d_out], vc.sum(0)) <= 1e-5 and _rel(stc[d_out:], (vc * vc).sum(0)) <= 2e-6


@pytest.mark.parametrize("c", [40, 47])
def test_fp32_module_runs_without_a_library_gemm(cuda, c):
    """The arxiv recipe (config 2: fp32, f = 128, d = 256, C = 40) through the module: every Linear (stems, GCN layers,
    attention projections are algebra on d x d, head) takes the streaming fp32 kernels — torch's matmul / addmm /
    F.linear are not called on [N, .] operands — and the result is the fp64 oracle's to the fp32 bar."""
    from oracle import sgformer_oracle as O
    from sgformer_amd import ops, synth
    from sgformer_amd.ours import SGFormer
    cfg = dict(synth.RECIPES["ogbn-arxiv"])
    n, f, d = 3000, 128, 256             # c = 47: the head's W / b are padded to 48 rows (ops.out_linear)
    torch.manual_seed(1)
    x, ei = torch.randn(n, f), O.synthetic_graph(n, 8.0, seed=2)
    y, idx = torch.randint(0, c, (n,)), torch.randperm(n)[: n // 2]
    p = O.init_params(cfg, f, d, c, seed=4)
    m = SGFormer(f, d, c, trans_dropout=0.0, gnn_dropout=0.0, **cfg)
    m.load_state_dict({**m.state_dict(), **p})
    m = m.to(cuda).train()
    big = []
    orig_linear, orig_addmm, orig_mm = torch.nn.functional.linear, torch.addmm, torch.Tensor.__matmul__

    def spy_linear(inp, *a, **k):
        if inp.dim() == 2 and inp.shape[0] == n:
            big.append(("F.linear", tuple(inp.shape)))
        return orig_linear(inp, *a, **k)

    def spy_matmul(self, other):
        if self.dim() == 2 and self.shape[0] == n:
            big.append(("matmul", tuple(self.shape)))
        return orig_mm(self, other)

    torch.nn.functional.linear, torch.Tensor.__matmul__ = spy_linear, spy_matmul
    try:
        logits = m(x.to(cuda), ei.to(cuda))
        O.nll_loss(logits, y.to(cuda), idx.to(cuda)).backward()
    finally:
        torch.nn.functional.linear, torch.Tensor.__matmul__ = orig_linear, orig_mm
    assert not big, big
    p64 = {k: v.double().requires_grad_(v.is_floating_point() and "running" not in k) for k, v in p.items()}
    ref = O.sgformer_forward(p64, x.double(), ei, cfg, training=True)
    O.nll_loss(ref, y, idx).backward()
    assert float((logits.detach().double().cpu() - ref.detach()).abs().max()) <= 1e-4
    gmax = max(float(v.grad.norm()) for v in p64.values() if v.grad is not None)
    for k, prm in m.named_parameters():
        g = p64[k].grad
        if g is not None:
            assert float((prm.grad.double().cpu() - g).norm()) <= 5e-4 * float(g.norm()) + 1e-6 * gmax, k


def test_gcn_epilogue_strided_operands(cuda):
    """Leading dimensions wider than the width: a column slice of a wider activation / weight matrix (the [. | x0]
    Linear of large/ours.py:36-38 takes W[:, :d] and W[:, d:])."""
    from sgformer_amd import ops
    g = torch.Generator().manual_seed(5)
    n, d = 333, 128
    a_wide = torch.randn(n, 2 * d, generator=g).bfloat16().to(cuda)
    w_wide = (torch.randn(d, 2 * d, generator=g) / d ** 0.5).bfloat16().to(cuda)
    a, w = a_wide[:, d:], w_wide[:, d:]
    y, _ = ops.K.gcn_epilogue_stats(a, w, None)
    ref = a.double() @ w.double().t()
    assert bool(((y.double() - ref).abs() <= 2.0 ** -8 * ref.abs() + 1e-6).all())
    dx = ops.K.gcn_epilogue_dx(a, w)
    refdx = a.double() @ w.double()
    assert bool(((dx.double() - refdx).abs() <= 2.0 ** -8 * refdx.abs() + 1e-6).all())


def test_gcn_epilogue_refusals(cuda):
    from sgformer_amd import ops
    from sgformer_amd._lib import SgfError
    assert not ops.K.gcn_epilogue_supported(100, 256, torch.bfloat16)
    assert ops.K.gcn_epilogue_supported(256, 256, torch.float32)          # fp32 storage: csrc/linear_f32.hip
    assert not ops.K.gcn_epilogue_supported(65, 256, torch.float32) and not ops.K.gcn_epilogue_supported(512, 256, torch.float32)
    assert not ops.K.gcn_epilogue_supported(512, 512, torch.bfloat16)
    a = torch.randn(10, 100, device=cuda).bfloat16()
    w = torch.randn(256, 100, device=cuda).bfloat16()
    with pytest.raises(SgfError):
        ops.K.gcn_epilogue_stats(a, w, None)
    a = torch.randn(10, 256, device=cuda).bfloat16()
    w = torch.randn(256, 256, device=cuda).bfloat16()
    with pytest.raises(SgfError):                       # rows not 16-byte aligned
        ops.K.gcn_epilogue_stats(a.view(-1)[4:4 + 9 * 256].view(9, 256), w, None)


@pytest.mark.parametrize("n,d", [(3000, 256), (517, 64)])
def test_linear_bn_stats_matches_unfused(cuda, n, d):
    """ops.linear_bn_stats (what GraphConv feeds its BatchNorm with) == ops.linear + ops.batch_stats, forward and
    all gradients, on the same inputs; the statistics against fp64 of the returned y."""
    from sgformer_amd import ops
    g = torch.Generator().manual_seed(n)
    x = torch.randn(n, d, generator=g).bfloat16().to(cuda)
    w = (torch.randn(d, d, generator=g) / d ** 0.5).to(cuda)
    b = (torch.randn(d, generator=g) * 0.1 + 0.5).to(cuda)
    go = torch.randn(n, d, generator=g).bfloat16().to(cuda)
    xa, wa, ba = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    y, (mean, var, n_tot) = ops.linear_bn_stats(xa, wa, ba)
    (y.float() * go.float()).sum().backward()
    yd = y.detach().double()
    assert n_tot == float(n)
    assert float((mean.double() - yd.mean(0)).abs().max()) <= 1e-5 * max(1.0, float(yd.abs().max()))
    assert _rel(var, yd.var(0, unbiased=False)) <= 1e-5
    ref = x.double() @ w.bfloat16().double().t() + b.double()
    assert bool(((yd - ref).abs() <= 2.0 ** -8 * ref.abs() + 1e-6).all())
    assert _rel(xa.grad.float(), go.double() @ w.bfloat16().double()) <= 4e-3
    assert _rel(wa.grad, go.double().t() @ x.double()) <= 2e-5
    assert _rel(ba.grad, go.double().sum(0)) <= 2e-5
    m2, v2, _ = ops.batch_stats(y.detach())
    assert float((m2 - mean).abs().max()) <= 1e-5 * max(1.0, float(yd.abs().max())) and _rel(var, v2) <= 1e-5


@pytest.mark.parametrize("n,d", [(1, 64), (33, 256), (255, 256), (1000, 128), (4096, 256), (20001, 256), (70003, 256), (9000, 64)])
def test_gcn_epilogue_cat_one_pass(cuda, n, d):
    """GraphConvLayer with use_init, large/ours.py:36-38: y = [a1 | a2] W^T + b in ONE pass (sgf_gcn_epilogue_cat: W resident
    for d <= 128, a PAIRED launch — two workgroups per row tile, one per half of the output columns — for d = 256).  Against
    fp64 on the host of the same bf16 operands: the sum of both products rounded ONCE; statistics = fp64 column sums of the
    returned y; run-to-run identical.  n covers fewer tiles than paired blocks, ragged last tiles, many tiles per wave."""
    from sgformer_amd import ops
    g = torch.Generator().manual_seed(11 * n + d)
    a1 = torch.randn(n, d, generator=g).bfloat16()
    a2 = torch.randn(n, d, generator=g).bfloat16()
    w = (torch.randn(d, 2 * d, generator=g) / (2 * d) ** 0.5).bfloat16()
    bias = torch.randn(d, generator=g)
    shift = torch.randn(d, generator=g) * 0.1
    y, st = ops.K.gcn_epilogue_cat(a1.to(cuda), a2.to(cuda), w.to(cuda), bias.to(cuda), shift.to(cuda), want_stats=True)
    ref = torch.cat([a1, a2], 1).double() @ w.double().t() + bias.double()
    err = (y.double().cpu() - ref).abs()
    assert bool((err <= 2.0 ** -8 * ref.abs() + 1e-6).all()), float((err - 2.0 ** -8 * ref.abs()).max())
    v = y.double().cpu() - shift.double()
    st_ref = torch.cat([v.sum(0), (v * v).sum(0)])
    tolst = 2e-6 * torch.cat([v.abs().sum(0), (v * v).sum(0)]).clamp_min(1e-3)
    assert bool(((st.double().cpu() - st_ref).abs() <= tolst).all())
    y2, none = ops.K.gcn_epilogue_cat(a1.to(cuda), a2.to(cuda), w.to(cuda), bias.to(cuda))
    assert none is None and torch.equal(y, y2)
    y3, st3 = ops.K.gcn_epilogue_cat(a1.to(cuda), a2.to(cuda), w.to(cuda), bias.to(cuda), shift.to(cuda), want_stats=True)
    assert torch.equal(y, y3) and torch.equal(st, st3)
    # strided operands (column slices of wider buffers), no bias / shift
    wide = torch.randn(n, 2 * d + 8, generator=g).bfloat16().to(cuda)
    b1, b2 = wide[:, :d], wide[:, d + 8:]
    y4, st4 = ops.K.gcn_epilogue_cat(b1, b2, w.to(cuda), None, None, want_stats=True)
    ref4 = torch.cat([b1, b2], 1).double().cpu() @ w.double().t()
    assert bool(((y4.double().cpu() - ref4).abs() <= 2.0 ** -8 * ref4.abs() + 1e-6).all())
    assert _rel(st4[:d], y4.double().sum(0)) <= 2e-6 or float(y4.double().sum(0).abs().max()) < 1e-3


@pytest.mark.parametrize("d,shifted,track", [(64, True, True), (256, False, True), (100, True, False), (257, True, True)])
def test_bn_finalize_matches_batchnorm_bookkeeping(cuda, d, shifted, track):
    """sgf_bn_finalize = what nn.BatchNorm1d does between its two passes (large/ours.py:87-88): batch mean / biased variance
    from the shifted sums, rstd = 1 / sqrt(var + eps), running_mean / running_var (unbiased) with momentum — against fp64."""
    from sgformer_amd import ops
    g = torch.Generator().manual_seed(d)
    n = 5000
    x = torch.randn(n, d, generator=g, dtype=torch.float64) * 1.7 + 0.4
    shift = x[:100].mean(0) if shifted else None
    v = x - (shift if shifted else 0.0)
    sums = torch.cat([v.sum(0), (v * v).sum(0)]).float().to(cuda)
    rm = torch.randn(d, generator=g).to(cuda)
    rv = (torch.rand(d, generator=g) + 0.5).to(cuda)
    rm0, rv0 = rm.double().cpu(), rv.double().cpu()
    mean, rstd = ops.K.bn_finalize(sums, shift.float().to(cuda) if shifted else None, float(n), 1e-5, 0.1,
                                   rm if track else None, rv if track else None)
    assert _rel(mean, x.mean(0)) <= 1e-6
    assert _rel(rstd, 1.0 / torch.sqrt(x.var(0, unbiased=False) + 1e-5)) <= 1e-5
    if track:
        assert _rel(rm, 0.9 * rm0 + 0.1 * x.mean(0)) <= 1e-6
        assert _rel(rv, 0.9 * rv0 + 0.1 * x.var(0, unbiased=True)) <= 1e-5
    else:
        assert torch.equal(rm.double().cpu(), rm0) and torch.equal(rv.double().cpu(), rv0)


@pytest.mark.parametrize("n,m,k", [(1, 64, 32), (100, 256, 100), (4099, 128, 128), (50001, 256, 100), (20000, 64, 64)])
@pytest.mark.parametrize("two", [False, True])
def test_gram_bn_bwd_without_dz(cuda, n, m, k, two):
    """sgf_bn_bwd_stats2 + sgf_gram_bn_bwd: dW / db of the GraphConv stem (large/ours.py:77-80 differentiated) straight from the
    one or two incoming gradients of its output — dz = BatchNorm'(relu'(g1 + g2)) is formed inside the Gram kernel's staging
    step and never written.  Against the explicit sequence on the same inputs: stats == sgf_bn_bwd_stats of the fp32 sum
    rounded... no: of g1 + g2 taken in fp32 (2e-6), dW == sgf_gram(sgf_bn_bwd_apply(g1 + g2), x) (2e-5: dz is rounded to
    bf16 once in both), db == its column sums."""
    from sgformer_amd import ops
    g = torch.Generator().manual_seed(3 * n + m + k + two)
    K = ops.K
    g1 = torch.randn(n, m, generator=g).bfloat16().to(cuda)
    g2 = torch.randn(n, m, generator=g).bfloat16().to(cuda) if two else None
    z = (torch.randn(n, m, generator=g) * 1.3 + 0.2).bfloat16().to(cuda)
    x = torch.randn(n, k, generator=g).bfloat16().to(cuda)
    mean = (torch.randn(m, generator=g) * 0.2 + 0.2).to(cuda)
    rstd = (1.0 / (1.0 + torch.rand(m, generator=g))).to(cuda)
    gamma = (1.0 + 0.3 * torch.randn(m, generator=g)).to(cuda)
    beta = (0.2 * torch.randn(m, generator=g)).to(cuda)
    gsum = g1.float() + (g2.float() if two else 0.0)
    stats = K.bn_bwd_stats2(g1, g2, z, mean, rstd, gamma, beta, True)
    xh = (z.double() - mean.double()) * rstd.double()
    gm = gsum.double() * ((xh * gamma.double() + beta.double()) > 0)
    ref_stats = torch.cat([gm.sum(0), (gm * xh).sum(0)])
    tol = 2e-6 * torch.cat([gm.abs().sum(0), (gm * xh).abs().sum(0)]).clamp_min(1e-3)
    assert bool(((stats.double() - ref_stats).abs() <= tol).all())
    inv_n = 1.0 / n
    dw, db = K.gram_bn_bwd(g1, g2, z, mean, rstd, gamma, beta, True, stats, inv_n, True, x)
    dzd = (gamma.double() * rstd.double() * (gm - stats[:m].double() * inv_n - xh * stats[m:].double() * inv_n))
    dz_r = dzd.float().bfloat16().double()                                   # one rounding, as the kernel applies it
    ref_dw = dz_r.t() @ x.double()
    assert _rel(dw, ref_dw) <= 2e-3 if n < 10 else _rel(dw, ref_dw) <= 3e-4, _rel(dw, ref_dw)
    assert _rel(db, dz_r.sum(0)) <= 3e-4 or float(dz_r.sum(0).abs().max()) < 1e-2
    dw2, db2 = K.gram_bn_bwd(g1, g2, z, mean, rstd, gamma, beta, True, stats, inv_n, True, x)
    assert torch.equal(dw, dw2) and torch.equal(db, db2)


@pytest.mark.parametrize("n,m,k,relu,affine", [(3, 256, 100, True, True), (1025, 256, 100, True, True),
                                               (4099, 128, 128, True, True), (50001, 256, 100, True, True),
                                               (30000, 64, 64, False, True), (20001, 256, 256, True, False),
                                               (9000, 64, 8, True, True)])
def test_gram_ln_bwd_without_dl(cuda, n, m, k, relu, affine):
    """sgf_gram_ln_bwd: dW / db of TransConv's stem Linear and d gamma / d beta of its LayerNorm (large/ours.py:198-201
    differentiated) from the gradient of the stem's output — dl = LayerNorm'(relu'(g)) is formed per patch inside the Gram
    kernel and never written.  Against fp64 on the same bf16 operands with dl rounded to bf16 once (as sgf_ln_bwd stores it):
    dW / db within 3e-4 (relative, Frobenius: the row means are fp32 sums in another order, which moves single roundings of dl),
    d gamma / d beta within 2e-6 of their absolute sums; and dW against the explicit sgf_ln_bwd + sgf_gram sequence."""
    from sgformer_amd import ops
    K = ops.K
    assert K.gram_ln_bwd_supported(m, k, torch.bfloat16)
    g = torch.Generator().manual_seed(5 * n + m + k)
    gr = torch.randn(n, m, generator=g).bfloat16().to(cuda)
    xin = (torch.randn(n, m, generator=g) * 1.7 + 0.3).bfloat16().to(cuda)
    x = torch.randn(n, k, generator=g).bfloat16().to(cuda)
    gamma = (1.0 + 0.3 * torch.randn(m, generator=g)).to(cuda) if affine else None
    beta = (0.2 * torch.randn(m, generator=g)).to(cuda) if affine else None
    if affine:
        h, mean, rstd = K.ln_fwd(xin, None, 1.0, 0.0, gamma, beta, relu, 1e-5)
    else:           # sgf_ln_fwd reads a null gamma as "no LayerNorm"; here null = LayerNorm without affine terms
        mean = xin.float().mean(1)
        rstd = (xin.float().var(1, unbiased=False) + 1e-5).rsqrt()
    dw, db, dg, dbt = K.gram_ln_bwd(gr, xin, mean, rstd, gamma, beta, relu, x)
    xh = (xin.double() - mean.double()[:, None]) * rstd.double()[:, None]
    ga = gamma.double() if affine else 1.0
    be = beta.double() if affine else 0.0
    gm = gr.double() * ((xh * ga + be) > 0) if relu else gr.double()
    dxh = gm * ga
    dl = rstd.double()[:, None] * (dxh - dxh.mean(1, keepdim=True) - xh * (dxh * xh).mean(1, keepdim=True))
    dl_r = dl.float().bfloat16().double()
    tol = 3e-3 if n < 10 else 3e-4
    assert _rel(dw, dl_r.t() @ x.double()) <= tol, _rel(dw, dl_r.t() @ x.double())
    assert _rel(db, dl_r.sum(0)) <= tol or float(dl_r.sum(0).abs().max()) < 1e-2
    if affine:
        assert bool(((dg.double() - (gm * xh).sum(0)).abs() <= 2e-6 * (gm * xh).abs().sum(0).clamp_min(1e-3)).all())
        assert bool(((dbt.double() - gm.sum(0)).abs() <= 2e-6 * gm.abs().sum(0).clamp_min(1e-3)).all())
    # the explicit sequence it replaces: dl written by sgf_ln_bwd, then sgf_gram
    if affine:
        dl_k = K.ln_bwd(gr, h if relu else None, xin, None, 1.0, 0.0, gamma, relu, mean, rstd)[0]
        dw_k, db_k = K.gram(dl_k, x, want_colsum=True)
        assert _rel(dw, dw_k) <= tol
    dw2, db2, dg2, dbt2 = K.gram_ln_bwd(gr, xin, mean, rstd, gamma, beta, relu, x)
    assert torch.equal(dw, dw2) and torch.equal(db, db2) and torch.equal(dg, dg2) and torch.equal(dbt, dbt2)


@pytest.mark.parametrize("n,m,k", [(100, 256, 256), (1025, 256, 256), (4099, 128, 128), (50001, 256, 256), (30000, 64, 64),
                                   (20000, 48, 256)])
def test_gram2_paired_launch(cuda, n, m, k):
    """sgf_gram2: dW = g^T [x_1 | x_2] of the two-operand Linear (large/ours.py:36-38 differentiated) from one PAIRED launch:
    bit-identical... to the fp32 sums of exact bf16 products in a different block order, so: both blocks within 2e-6 (relative,
    Frobenius) of fp64 on the same bf16 operands, the column sums likewise, results written into column slices of one
    matrix, and run-to-run identical."""
    from sgformer_amd import ops
    g = torch.Generator().manual_seed(n + m + k)
    a = torch.randn(n, m, generator=g).bfloat16().to(cuda)
    b1 = torch.randn(n, k, generator=g).bfloat16().to(cuda)
    b2 = torch.randn(n, k, generator=g).bfloat16().to(cuda)
    dw = torch.empty(m, 2 * k, device=cuda)
    cs = ops.K.gram2(a, b1, b2, dw[:, :k], dw[:, k:], want_colsum=True)
    ad = a.double().cpu()
    assert _rel(dw[:, :k], ad.t() @ b1.double().cpu()) <= 2e-6
    assert _rel(dw[:, k:], ad.t() @ b2.double().cpu()) <= 2e-6
    assert _rel(cs, ad.sum(0)) <= 2e-6 or float(ad.sum(0).abs().max()) < 1e-2
    dw2 = torch.empty_like(dw)
    cs2 = ops.K.gram2(a, b1, b2, dw2[:, :k], dw2[:, k:], want_colsum=True)
    assert torch.equal(dw, dw2) and torch.equal(cs, cs2)
    one, _ = ops.K.gram(a, b1, want_colsum=False)
    assert _rel(dw[:, :k], one) <= 1e-6


@pytest.mark.parametrize("n,d", [(1, 64), (100, 256), (31, 128), (2049, 64), (4096, 256), (9001, 128), (20001, 256), (70003, 256)])
@pytest.mark.parametrize("relu,training", [(True, True), (False, True), (True, False)])
def test_gcn_bn_bwd_dx_chain(cuda, n, d, relu, training):
    """sgf_gcn_bn_bwd_dx, three chained calls as the three GraphConv layers of the products recipe issue them in their
    backward (large/ours.py:83-93 differentiated): per call dz bit-equal (or within one bf16 ulp on a vanishing fraction:
    the two kernels contract their multiply-adds alike) to sgf_bn_bwd_apply, d y = dz W[:, :d] against fp64 of the rounded
    dz, and the accumulated gradient of x0 = sum_i (gy_i + dz_i W_i[:, d:]) against fp64 (three bf16 roundings of the
    running sum).  n covers: fewer tiles than launched blocks, ragged last tiles, several tiles per wave."""
    from sgformer_amd import ops
    g = torch.Generator().manual_seed(5 * n + d + relu)
    K = ops.K
    acc, ref_acc = None, torch.zeros(n, d, dtype=torch.float64)
    for layer in range(3):
        gy = torch.randn(n, d, generator=g).bfloat16().to(cuda)
        z = (torch.randn(n, d, generator=g) * 1.5 + 0.3).bfloat16().to(cuda)
        mean = (torch.randn(d, generator=g) * 0.2 + 0.3).to(cuda)
        rstd = (1.0 / (1.0 + torch.rand(d, generator=g))).to(cuda)
        gamma = (1.0 + 0.3 * torch.randn(d, generator=g)).to(cuda)
        beta = (0.2 * torch.randn(d, generator=g)).to(cuda)
        w = (torch.randn(d, 2 * d, generator=g) / (2 * d) ** 0.5).bfloat16().to(cuda)
        stats = K.bn_bwd_stats(gy, z, mean, rstd, gamma, beta, relu)
        inv_n = 1.0 / n
        dz_ref = K.bn_bwd_apply(gy, z, mean, rstd, gamma, beta, relu, stats, inv_n, training)
        add_gy = layer != 1
        dz, dy, acc = K.gcn_bn_bwd_dx(gy, z, mean, rstd, gamma, beta, relu, stats, inv_n, training, w, acc,
                                      last=(layer == 2), add_gy=add_gy)
        differ = int((dz != dz_ref).sum())
        assert differ <= max(2, int(1e-3 * dz.numel())), differ
        # (the fused kernel evaluates the same expression with its coefficients folded — cs g' + z Z1 + Z0 — : fp32 rounding
        # differs, so a result next to a bf16 rounding boundary may land on the other side, and a cancelling one moves by
        # ~1e-6 of the operands' size)
        ulp = 2.0 ** -7 * dz_ref.float().abs() + 2e-5
        assert bool(((dz.float() - dz_ref.float()).abs() <= ulp).all())
        dzd = dz.double().cpu()
        ref_dy = dzd @ w[:, :d].double().cpu()
        err = (dy.double().cpu() - ref_dy).abs()
        assert bool((err <= 2.0 ** -8 * ref_dy.abs() + 1e-6).all()), float(err.max())
        ref_acc = ref_acc + dzd @ w[:, d:].double().cpu() + (gy.double().cpu() if add_gy else 0.0)
    assert acc.shape == (n, d) and acc.dtype == torch.bfloat16
    err = (acc.double().cpu() - ref_acc).abs()
    scale = ref_acc.abs() + 1.0
    assert bool((err <= 6 * 2.0 ** -8 * scale).all()), float((err / scale).max())   # 3 roundings of running sums > total


def test_gcn_layers_fused_vs_unfused(cuda, monkeypatch):
    """GraphConv (products recipe: 3 layers, use_init, residual, BatchNorm, bf16) with the layers as ONE autograd node each
    (ops.linear_bn_act_res: sgf_gcn_epilogue_cat forward, sgf_gcn_bn_bwd_dx backward, layer_[0]'s gradient accumulated in
    the kernels) against the same module with SGF_GCN_FUSED=0 (separate Linear / BatchNorm nodes + the fan-out hub): same
    outputs up to the single rounding of the two-operand Linear, same gradients to bf16 accuracy."""
    from sgformer_amd import ops
    from sgformer_amd.ours import GraphConv
    from oracle import sgformer_oracle as O
    n, f, d = 5000, 32, 128
    torch.manual_seed(0)
    ei = O.synthetic_graph(n, 8.0, seed=3).to(cuda)
    x = torch.randn(n, f).bfloat16().to(cuda)
    go = torch.randn(n, d).bfloat16().to(cuda)
    res = {}
    for mode in ("1", "0", "bwd"):
        monkeypatch.setenv("SGF_GCN_FUSED", "1" if mode == "bwd" else mode)
        monkeypatch.setenv("SGF_GCN_BWD_FUSED", "1" if mode == "bwd" else "0")
        torch.manual_seed(1)
        m = GraphConv(f, d, num_layers=3, dropout=0.0, use_bn=True, use_residual=True, use_weight=True, use_init=True,
                      use_act=True).to(cuda).train()
        out = m(x, ei)
        (out.float() * go.float()).sum().backward()
        res[mode] = (out.detach().float(), {k: p.grad.detach().float() for k, p in m.named_parameters()},
                     {k: b.detach().float().clone() for k, b in m.named_buffers()})
    o0, g0, b0 = res["0"]
    for mode in ("1", "bwd"):          # one node per layer with the separate backward kernels / with sgf_gcn_bn_bwd_dx
        o1, g1, b1 = res[mode]
        assert _rel(o1, o0) <= 1e-2
        for k in g0:
            if k.endswith("W.bias") or k == "fcs.0.bias":
                # a Linear bias in front of a BatchNorm has the exact gradient 0 (the mean is removed): both are rounding noise
                # (a column sum of ~5000 bf16-rounded dz values: the roundings do not cancel)
                assert float(g1[k].abs().max()) <= 1e-1 * max(1.0, float(g0["bns.1.bias"].abs().max())), k
                continue
            assert _rel(g1[k], g0[k]) <= 3e-2, (mode, k, _rel(g1[k], g0[k]))
        for k in b0:
            assert _rel(b1[k], b0[k]) <= 1e-3 or "num_batches" in k, (mode, k)


@pytest.mark.parametrize("n,d", [(1, 64), (33, 256), (1000, 128), (20001, 256)])
def test_gcn_epilogue_two_operands(cuda, n, d, monkeypatch):
    """The two-pass form (SGF_GCN_CAT=0; also what fp32 storage runs): y = [a1 | a2] W^T + b in two streaming passes with
    the first product kept in the accumulator layout.  Reference on the host in fp64: first product rounded to bf16 (what
    the partial buffer stores), the sum rounded once; statistics of the returned y; both input gradients."""
    from sgformer_amd import ops
    monkeypatch.setenv("SGF_GCN_CAT", "0")
    g = torch.Generator().manual_seed(7 * n + d)
    a1 = torch.randn(n, d, generator=g).bfloat16()
    a2 = torch.randn(n, d, generator=g).bfloat16()
    w = (torch.randn(d, 2 * d, generator=g) / (2 * d) ** 0.5).bfloat16()
    bias = torch.randn(d, generator=g)
    shift = torch.randn(d, generator=g) * 0.1
    y, st = ops.K.gcn_epilogue_cat(a1.to(cuda), a2.to(cuda), w.to(cuda), bias.to(cuda), shift.to(cuda), want_stats=True)
    part = (a1.double() @ w[:, :d].double().t() + bias.double())
    ref = a2.double() @ w[:, d:].double().t() + part
    # the rounded partial may sit one bf16 ulp of `part` away from the exact one; then one rounding of the sum
    tol = 2.0 ** -8 * (ref.abs() + part.abs()) + 1e-6
    assert bool(((y.double().cpu() - ref).abs() <= tol).all())
    part_r = part.float().bfloat16().double()
    ref_r = a2.double() @ w[:, d:].double().t() + part_r
    frac_exact = float(((y.double().cpu() - ref_r).abs() <= 2.0 ** -8 * ref_r.abs() + 1e-6).float().mean())
    assert frac_exact >= 0.999, frac_exact            # single rounding of (product 2 + rounded product 1)
    v = y.double().cpu() - shift.double()
    st_ref = torch.cat([v.sum(0), (v * v).sum(0)])
    tolst = 2e-6 * torch.cat([v.abs().sum(0), (v * v).sum(0)]).clamp_min(1e-3)
    assert bool(((st.double().cpu() - st_ref).abs() <= tolst).all())
    y2, none = ops.K.gcn_epilogue_cat(a1.to(cuda), a2.to(cuda), w.to(cuda), bias.to(cuda))
    assert none is None and torch.equal(y, y2)


def test_linear_cat_bn_stats_matches_unfused(cuda):
    """ops.linear_bn_stats((y, x0), W, b) == addmm path + batch_stats: values, statistics and all gradients."""
    from sgformer_amd import ops
    g = torch.Generator().manual_seed(3)
    n, d = 2500, 256
    x1 = torch.randn(n, d, generator=g).bfloat16().to(cuda)
    x2 = torch.randn(n, d, generator=g).bfloat16().to(cuda)
    w = (torch.randn(d, 2 * d, generator=g) / (2 * d) ** 0.5).to(cuda)
    b = (torch.randn(d, generator=g) * 0.1).to(cuda)
    go = torch.randn(n, d, generator=g).bfloat16().to(cuda)
    a1, a2 = x1.clone().requires_grad_(True), x2.clone().requires_grad_(True)
    wa, ba = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    y, (mean, var, n_tot) = ops.linear_bn_stats((a1, a2), wa, ba)
    (y.float() * go.float()).sum().backward()
    yd = y.detach().double()
    wr = w.bfloat16().double()
    ref = torch.cat([x1, x2], 1).double() @ wr.t() + b.double()
    part = x1.double() @ wr[:, :d].t() + b.double()              # stored rounded: its ulp enters the sum's error
    assert bool(((yd - ref).abs() <= 2.0 ** -8 * (ref.abs() + part.abs()) + 1e-6).all())
    assert float((mean.double() - yd.mean(0)).abs().max()) <= 1e-5 * max(1.0, float(yd.abs().max()))
    assert _rel(var, yd.var(0, unbiased=False)) <= 1e-5
    assert _rel(a1.grad.float(), go.double() @ wr[:, :d]) <= 4e-3
    assert _rel(a2.grad.float(), go.double() @ wr[:, d:]) <= 4e-3
    assert _rel(wa.grad, go.double().t() @ torch.cat([x1, x2], 1).double()) <= 2e-5
    assert _rel(ba.grad, go.double().sum(0)) <= 2e-5


@pytest.mark.parametrize("n,d", [(1, 64), (77, 128), (3001, 256), (20000, 256)])
def test_attn_h_backward_split_form(cuda, n, d):
    """sgf_attn_h_bwd_pre -> _reduce_scaled -> _post (bf16): the backward of large/ours.py:130-151 for one head from the
    un-projected input, against fp64 ON THE HOST of the same bf16 inputs: hstats = [h^T dnum | h^T dden | sum dnum | sum dden]
    relative 2e-3 on the matrix (dnum is re-rounded to bf16 for the matrix cores) and 1e-5 on the vectors the kernel keeps
    in fp32; dh within bf16 rounding (of the result and of the parked first product); and the row scalars themselves."""
    from sgformer_amd import ops
    g_ = torch.Generator().manual_seed(n + d)
    h = torch.randn(n, d, generator=g_).bfloat16()
    g = torch.randn(n, d, generator=g_).bfloat16()
    M = torch.randn(d, d, generator=g_) / d ** 0.5
    D = torch.randn(d, d, generator=g_) / d ** 0.5
    m, w, ds = torch.randn(d, generator=g_), torch.rand(d, generator=g_) / d, torch.randn(d, generator=g_)
    beta = torch.full((1,), 3.0)
    K = ops.K
    hc, gc = h.to(cuda), g.to(cuda)
    out, den = K.attn_h_fwd(hc, M.to(cuda), m.to(cuda), w.to(cuda), beta.to(cuda))
    assert K.attn_h_bwd_split_supported(hc, gc, out)
    # forward: den = h.w + beta to fp32 accuracy (w enters the matrix cores as hi + lo halves), out within bf16 rounding
    den_ref = h.double() @ w.double() + 3.0
    assert float(((den.double().cpu().reshape(-1) - den_ref).abs() / den_ref.abs()).max()) <= 2e-5
    out_ref = (h.double() @ M.bfloat16().double() + m.double()) / den_ref.reshape(-1, 1)
    assert bool(((out.double().cpu() - out_ref).abs() <= 2.0 ** -8 * out_ref.abs() + 1e-5).all())
    rowscal = K.attn_h_bwd_pre(gc, out, den, M.to(cuda), w.to(cuda))
    hstats = K.attn_h_bwd_reduce_scaled(hc, gc, rowscal)
    dh = K.attn_h_bwd_post(hc, D.to(cuda), ds.to(cuda))
    hd, gd, od, dend = h.double(), g.double(), out.double().cpu(), den.double().cpu().reshape(-1, 1)
    inv = 1.0 / dend
    dden = -(gd * od).sum(1, keepdim=True) * inv
    assert _rel(rowscal[:, 0], inv.reshape(-1)) <= 1e-6 and _rel(rowscal[:, 1], dden.reshape(-1)) <= 1e-5
    dnum = gd * inv
    assert _rel(hstats[:d * d].reshape(d, d), hd.t() @ dnum) <= 2e-3
    assert _rel(hstats[d * d:d * d + d], (hd * dden).sum(0)) <= 1e-5
    assert _rel(hstats[d * d + d:d * d + 2 * d], dnum.sum(0)) <= 1e-5
    assert abs(float(hstats[-1]) - float(dden.sum())) <= 1e-5 * max(1.0, float(dden.abs().sum()))
    Mr, Dr = M.bfloat16().double(), D.bfloat16().double()        # the resident matrices are rounded to bf16 (as in r01)
    part = dnum @ Mr.t() + dden * w.double()
    ref = part + hd @ Dr + ds.double()
    err = (dh.double().cpu() - ref).abs()
    assert bool((err <= 2.0 ** -8 * (ref.abs() + part.abs()) + 1e-4).all()), float(err.max())
    # and the one-call forms agree with the three-call form
    hs_old = K.attn_h_bwd_reduce(hc, gc, out, den)
    assert _rel(hs_old, hstats) <= 2e-3
    dh_old = K.attn_h_bwd_apply(hc, gc, out, den, M.to(cuda), w.to(cuda), D.to(cuda), ds.to(cuda))
    assert torch.equal(dh_old, dh)


@pytest.mark.parametrize("n,f,d", [(1, 100, 64), (77, 100, 256), (3001, 128, 128), (1000, 64, 256), (20001, 100, 256),
                                   (500, 4, 64)])
def test_stem_pair(cuda, n, f, d):
    """K10, large/ours.py:77 and :198: both branches' first Linear from one read of x (rows of f bf16 elements, 8-byte
    aligned only), against fp64 ON THE HOST of the same bf16 operands; the first output's BatchNorm sums against fp64
    sums of the returned tensor; one-output form; gradients through ops.stem_pair against fp64."""
    from sgformer_amd import ops
    g = torch.Generator().manual_seed(n + f + d)
    x = torch.randn(n, f, generator=g).bfloat16()
    w0 = (torch.randn(d, f, generator=g) / f ** 0.5)
    w1 = (torch.randn(d, f, generator=g) / f ** 0.5)
    b0, b1 = torch.randn(d, generator=g), torch.randn(d, generator=g)
    shift = torch.randn(d, generator=g) * 0.1
    assert ops.K.stem_pair_supported(f, d, torch.bfloat16)
    xc = x.to(cuda)
    w0c, w1c = w0.bfloat16().to(cuda), w1.bfloat16().to(cuda)
    y0, y1, st = ops.K.stem_pair(xc, w0c, b0.to(cuda), w1c, b1.to(cuda), shift.to(cuda), want_stats0=True)
    r0 = x.double() @ w0.bfloat16().double().t() + b0.double()
    r1 = x.double() @ w1.bfloat16().double().t() + b1.double()
    assert bool(((y0.double().cpu() - r0).abs() <= 2.0 ** -8 * r0.abs() + 1e-6).all())
    assert bool(((y1.double().cpu() - r1).abs() <= 2.0 ** -8 * r1.abs() + 1e-6).all())
    v = y0.double().cpu() - shift.double()
    st_ref = torch.cat([v.sum(0), (v * v).sum(0)])
    tol = 2e-6 * torch.cat([v.abs().sum(0), (v * v).sum(0)]).clamp_min(1e-3)
    assert bool(((st.double().cpu() - st_ref).abs() <= tol).all())
    ya, none1, none2 = ops.K.stem_pair(xc, w0c, None, None, None)
    ra = x.double() @ w0.bfloat16().double().t()
    assert none1 is None and none2 is None
    assert bool(((ya.double().cpu() - ra).abs() <= 2.0 ** -8 * ra.abs() + 1e-6).all())
    # autograd form
    w0g, w1g = w0.to(cuda).requires_grad_(True), w1.to(cuda).requires_grad_(True)
    b0g, b1g = b0.to(cuda).requires_grad_(True), b1.to(cuda).requires_grad_(True)
    go0 = torch.randn(n, d, generator=g).bfloat16()
    go1 = torch.randn(n, d, generator=g).bfloat16()
    assert ops.stem_pair_supported(xc, w0g, w1g)
    (z0, z1), stats = ops.stem_pair(xc, w0g, b0g, w1g, b1g, want_stats0=True)
    (z0.float() * go0.to(cuda).float()).sum().backward(retain_graph=True)
    (z1.float() * go1.to(cuda).float()).sum().backward()
    assert torch.equal(z0, y0) and torch.equal(z1, y1)
    mean, var, cnt = stats
    yd = y0.double().cpu()
    assert cnt == float(n) and float((mean.double().cpu() - yd.mean(0)).abs().max()) <= 1e-5 * max(1.0, float(yd.abs().max()))
    if n > 1:
        assert _rel(var, yd.var(0, unbiased=False)) <= 1e-5
    assert _rel(w0g.grad, go0.double().t() @ x.double()) <= 2e-5 and _rel(w1g.grad, go1.double().t() @ x.double()) <= 2e-5
    assert _rel(b0g.grad, go0.double().sum(0)) <= 2e-5 and _rel(b1g.grad, go1.double().sum(0)) <= 2e-5


def test_attn_h_bwd_post_addend(cuda):
    """sgf_attn_h_bwd_post with a second gradient folded in == the un-folded result + that gradient, rounded once more
    (what autograd's add of the two bf16 gradients gives)."""
    from sgformer_amd import ops
    g_ = torch.Generator().manual_seed(9)
    n, d = 4100, 256
    h = torch.randn(n, d, generator=g_).bfloat16().to(cuda)
    g = torch.randn(n, d, generator=g_).bfloat16().to(cuda)
    extra = torch.randn(n, d, generator=g_).bfloat16().to(cuda)
    M = (torch.randn(d, d, generator=g_) / 16).to(cuda)
    D = (torch.randn(d, d, generator=g_) / 16).to(cuda)
    m, w, ds = (torch.randn(d, generator=g_).to(cuda), (torch.rand(d, generator=g_) / d).to(cuda),
                torch.randn(d, generator=g_).to(cuda))
    beta = torch.full((1,), 3.0, device=cuda)
    K = ops.K
    out, den = K.attn_h_fwd(h, M, m, w, beta)
    K.attn_h_bwd_pre(g, out, den, M, w)
    plain = K.attn_h_bwd_post(h, D, ds)
    folded = K.attn_h_bwd_post(h, D, ds, extra)
    assert torch.equal(folded, (plain.float() + extra.float()).bfloat16())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_trainer_loss_lines_in_one_pass(cuda, dtype):
    """large/main.py:139-141 as written, under the launcher's patches, on the device: F.log_softmax -> lazy tensor,
    `out[train_idx]` -> lazy rows, nn.NLLLoss() -> sgf_nll_fwd / sgf_nll_bwd on the logits.  Against ATen's three lines on the
    same logits in fp32: loss 1e-6, gradient 1e-6 (bf16 logits: one bf16 rounding of the gradient); rows whose target is
    ignore_index leave the mean and get a ZERO gradient row."""
    import torch.nn as nn
    import torch.nn.functional as F
    from sgformer_amd import launch, ops
    from sgformer_amd.loss import LazyLogSoftmax
    g = torch.Generator().manual_seed(11)
    n, c = 50_000, 47
    logits = (torch.randn(n, c, generator=g) * 2).to(dtype).to(cuda).requires_grad_(True)
    label = torch.randint(0, c, (n, 1), generator=g).to(cuda)
    idx = torch.randperm(n, generator=g)[:20_000].to(cuda)
    tgt = label.squeeze(1)[idx].clone()
    tgt[::7] = -100
    ref_in = logits.detach().float().requires_grad_(True)
    ref = F.nll_loss(F.log_softmax(ref_in, dim=1)[idx], tgt)
    g_ref, = torch.autograd.grad(ref, ref_in)
    launch.patch_nll_loss()
    try:
        calls = []
        real = ops.K.nll_fwd
        ops.K.nll_fwd = staticmethod(lambda *a: (calls.append(1), real(*a))[1])
        full = []
        value = LazyLogSoftmax._sgf_value
        LazyLogSoftmax._sgf_value = lambda self: (full.append(1), value(self))[1]
        out = F.log_softmax(logits, dim=1)
        loss = nn.NLLLoss()(out[idx], tgt)
        assert isinstance(out, LazyLogSoftmax) and calls == [1] and out.shape == (n, c)
        assert full == [], "the one-pass path must not compute the full log-softmax"
        g_got, = torch.autograd.grad(loss, logits)
        # large/main-batch.py:146: the rows are picked by a BOOLEAN mask that lives on the host
        mask = torch.zeros(n, dtype=torch.bool)
        mask[idx.cpu()] = True
        t_m = label.squeeze(1)[mask]
        loss_m = nn.NLLLoss()(F.log_softmax(logits, dim=1)[mask], t_m)
        assert calls == [1, 1] and full == []
        g_m, = torch.autograd.grad(loss_m, logits)
    finally:
        ops.K.nll_fwd = real
        LazyLogSoftmax._sgf_value = value
        launch.unpatch_nll_loss()
    ref_m = F.nll_loss(F.log_softmax(ref_in, dim=1)[mask], t_m)
    g_ref_m, = torch.autograd.grad(ref_m, ref_in)
    assert abs(float(loss_m) - float(ref_m)) <= 1e-6 * max(1.0, abs(float(ref_m)))
    assert float((g_m.float() - g_ref_m).abs().max()) <= (1e-6 if dtype == torch.float32 else 2.0 ** -8) * float(g_ref_m.abs().max())
    assert abs(float(loss) - float(ref)) <= 1e-6 * max(1.0, abs(float(ref)))
    tol = 1e-6 if dtype == torch.float32 else 2.0 ** -8
    assert float((g_got.float() - g_ref).abs().max()) <= tol * float(g_ref.abs().max())
    ignored = idx[::7]
    assert float(g_got[ignored].abs().max()) == 0.0


@pytest.mark.parametrize("n,d,c", [(1000, 256, 47), (33, 128, 7), (5001, 256, 64), (2500, 64, 2)])
def test_combine_fc_row_mapped(cuda, n, d, c):
    """sgf_combine_fc_fwd_mapped / _bwd_mapped: the fused head with the module's row permutation folded into its stores
    (forward: row j of the product lands in row row_map[j]) and loads (backward: row j of dx reads row row_map[j] of the
    incoming gradient) — bit-identical to the unmapped kernels followed / preceded by the explicit permutation."""
    from sgformer_amd import ops
    g = torch.Generator().manual_seed(n + d + c)
    x1 = torch.randn(n, d, generator=g).bfloat16().to(cuda)
    x2 = torch.randn(n, d, generator=g).bfloat16().to(cuda)
    w = (torch.randn(c, d, generator=g) / d ** 0.5).to(cuda)
    b = (torch.randn(c, generator=g) * 0.1).to(cuda)
    go = torch.randn(n, c, generator=g).to(cuda)
    perm = torch.randperm(n, generator=g).to(torch.int32).to(cuda)
    assert ops.combine_fc_mapped_supported(x1, c)
    plain = ops.K.combine_fc_fwd(x1, 0.8, x2, 0.2, w, b)
    mapped = ops.K.combine_fc_fwd(x1, 0.8, x2, 0.2, w, b, perm)
    want = torch.empty_like(plain)
    want[perm.long()] = plain
    assert torch.equal(mapped, want)
    d1, d2 = ops.K.combine_fc_bwd(go[perm.long()].contiguous(), w, 0.8, 0.2, torch.bfloat16)
    m1, m2 = ops.K.combine_fc_bwd(go, w, 0.8, 0.2, torch.bfloat16, perm)
    assert torch.equal(d1, m1) and torch.equal(d2, m2)
    # the autograd node: same logits / gradients as the unmapped node between two explicit permutations
    leaves = [t.clone().requires_grad_(True) for t in (x1, x2, w, b)]
    out = ops.combine_fc(leaves[1], leaves[0], leaves[2], leaves[3], 0.8, 0.2, perm)
    (out * go).sum().backward()
    ref_l = [t.clone().requires_grad_(True) for t in (x1, x2, w, b)]
    ref = ops.combine_fc(ref_l[1], ref_l[0], ref_l[2], ref_l[3], 0.8, 0.2)
    inv = torch.empty_like(perm)
    inv[perm.long()] = torch.arange(n, dtype=torch.int32, device=cuda)
    (ref[inv.long()] * go).sum().backward()
    assert torch.equal(out, ref[inv.long()])
    for a, r in zip(leaves, ref_l):
        assert _rel(a.grad.float(), r.grad.float()) <= 1e-6


@pytest.mark.parametrize("n,d", [(1, 64), (100, 256), (31, 128), (4096, 256), (9001, 128), (20001, 256), (70003, 256), (5000, 64)])
@pytest.mark.parametrize("with_g,with_acc", [(True, True), (True, False), (False, True), (False, False)])
def test_gcn_epilogue_dx2_acc(cuda, n, d, with_g, with_acc):
    """sgf_gcn_epilogue_dx2_acc: dy = dz W[:, :d] and acc_out = dz W[:, d:] + gadd + acc_in from one launch (balanced pairs of
    workgroups at d = 256).  dy bit-identical to sgf_gcn_epilogue_dx; acc_out == bf16(fp32(bf16(dz W2)) + gadd + acc_in):
    the product is rounded once on its way through the staging patch (as the separate kernel stores it), the sum once more —
    compared with exactly that arithmetic built from sgf_gcn_epilogue_dx's output (equal up to the ties of the last
    rounding: <= 1 bf16 ulp on <= 1e-3 of the elements), and with fp64 within 2 bf16 roundings."""
    from sgformer_amd import ops
    K = ops.K
    assert K.gcn_epilogue_dx2_acc_supported(d, torch.bfloat16)
    g = torch.Generator().manual_seed(7 * n + d + 2 * with_g + with_acc)
    dz = torch.randn(n, d, generator=g).bfloat16().to(cuda)
    w = (torch.randn(d, 2 * d, generator=g) / (2 * d) ** 0.5).bfloat16().to(cuda)
    gadd = torch.randn(n, d, generator=g).bfloat16().to(cuda) if with_g else None
    acc_in = (torch.randn(n, d, generator=g) * 2).bfloat16().to(cuda) if with_acc else None
    dy, acc = K.gcn_epilogue_dx2_acc(dz, w, gadd, acc_in)
    dy_ref = K.gcn_epilogue_dx(dz, w[:, :d])
    p2 = K.gcn_epilogue_dx(dz, w[:, d:])
    assert torch.equal(dy, dy_ref)
    want = p2.float()
    if with_g:
        want = want + gadd.float()
    if with_acc:
        want = want + acc_in.float()
    want_b = want.bfloat16()
    diff = (acc.float() - want_b.float()).abs()
    ulp = want_b.float().abs().clamp_min(1e-30) * 2.0 ** -7
    assert bool((diff <= ulp).all()) and int((diff > 0).sum()) <= max(2, int(1e-3 * acc.numel()))
    ref64 = dz.double() @ w[:, d:].double()
    if with_g:
        ref64 = ref64 + gadd.double()
    if with_acc:
        ref64 = ref64 + acc_in.double()
    assert float((acc.double() - ref64).abs().max()) <= 2.0 ** -7 * float(ref64.abs().max()) + 1e-6
    dy2, acc2 = K.gcn_epilogue_dx2_acc(dz, w, gadd, acc_in)
    assert torch.equal(dy, dy2) and torch.equal(acc, acc2)
