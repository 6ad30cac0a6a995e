"""r05: the general matrix-core Linear (sgf_gemm), the entry padding (sgf_pad_rows) and the attention's d x d algebra inside
libsgf.so (sgf_attn_h_small_fwd / _bwd) against fp64 restatements — and the proof that NO recipe shape reaches a library
GEMM any more: the device kernels of a pokec (f = 65), a Cora (f = 1433, medium/ours.py) and a multi-head step are listed
with the profiler and must not contain a Tensile / rocBLAS / hipBLASLt kernel."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import sgformer_oracle as O  # noqa: E402


@pytest.fixture(scope="module")
def cuda():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def _rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-300))


# ------------------------------------------------------------------------------------------------
# sgf_gemm
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("m,n,k", [(1, 1, 1), (63, 65, 31), (64, 64, 32), (200, 7, 1433), (333, 257, 514), (1000, 47, 65),
                                   (129, 300, 33), (5, 513, 2)])
@pytest.mark.parametrize("adt,bdt,cdt", [(torch.float32, torch.float32, torch.float32),
                                         (torch.bfloat16, torch.bfloat16, torch.bfloat16),
                                         (torch.bfloat16, torch.bfloat16, torch.float32),
                                         (torch.bfloat16, torch.float32, torch.float32)])
def test_gemm_matches_fp64(cuda, m, n, k, adt, bdt, cdt):
    """c = a b + bias for arbitrary sizes and every storage combination: fp32 operands on the exact-fp32 matrix cores to
    summation-order accuracy, bf16 products exact with fp32 sums, one rounding into the output dtype."""
    from sgformer_amd import ops
    g = torch.Generator().manual_seed(m * 7 + n * 3 + k)
    a = torch.randn(m, k, generator=g).to(adt).to(cuda)
    b = (torch.randn(k, n, generator=g) / max(k, 1) ** 0.5).to(bdt).to(cuda)
    bias = torch.randn(n, generator=g).to(cuda)
    out = ops.K.gemm(a, b, bias=bias, out_dtype=cdt)
    ref = a.double() @ b.double() + bias.double()
    assert out.dtype == cdt and out.shape == (m, n)
    rt, at = (2.0 ** -8, 1e-5) if cdt == torch.bfloat16 else (0.0, 5e-6)
    assert bool(((out.double() - ref).abs() <= rt * ref.abs() + at * max(1.0, float(ref.abs().max()))).all())


def test_gemm_strided_views_and_accumulate(cuda):
    """Transposed / column-sliced operands are passed by their strides (y = x W^T: b = W.t(); dx = g W[:, blk]), alpha
    from a device scalar, beta * addend with the addend aliasing the output."""
    from sgformer_amd import ops
    g = torch.Generator().manual_seed(11)
    x = torch.randn(301, 70, generator=g).to(cuda)
    w = (torch.randn(50, 140, generator=g) / 8).to(cuda)
    y = ops.K.gemm(x, w[:, 70:].t())                                          # x W2^T
    assert _rel(y, x.double() @ w[:, 70:].double().t()) <= 2e-6
    gy = torch.randn(301, 50, generator=g).to(cuda)
    dx = ops.K.gemm(gy, w[:, :70])                                            # g W1
    assert _rel(dx, gy.double() @ w[:, :70].double()) <= 2e-6
    at = ops.K.gemm(x.t(), gy)                                                # x^T g: the A operand walked along its rows
    assert _rel(at, x.double().t() @ gy.double()) <= 2e-6
    c = torch.full((1,), 0.37, device=cuda)
    acc = y.clone()
    ops.K.gemm(x, w[:, :70].t(), out=acc, alpha=2.0, alpha_dev=c, beta=0.5, addend=acc)
    ref = 2.0 * 0.37 * (x.double() @ w[:, :70].double().t()) + 0.5 * y.double()
    assert _rel(acc, ref) <= 2e-6
    wide = torch.zeros(301, 64, device=cuda)                                   # output with a leading dimension > n
    ops.K.gemm(x, w[:, :70].t(), out=wide[:, 7:57])
    assert _rel(wide[:, 7:57], x.double() @ w[:, :70].double().t()) <= 2e-6 and float(wide[:, :7].abs().max()) == 0.0
    assert ops.K.gemm(x[:0], w[:, :70].t()).shape == (0, 50)


def test_gemm_is_deterministic_and_exact_on_small_integers(cuda):
    from sgformer_amd import ops
    g = torch.Generator().manual_seed(3)
    a = torch.randint(-8, 9, (257, 130), generator=g).float().to(cuda)
    b = torch.randint(-8, 9, (130, 77), generator=g).float().to(cuda)
    out = ops.K.gemm(a, b)
    assert bool((out.double() == a.double() @ b.double()).all())              # exact FMA chain: integers stay exact
    assert bool((ops.K.gemm(a, b) == out).all())
    ob = ops.K.gemm(a.bfloat16(), b.bfloat16(), out_dtype=torch.float32)
    assert bool((ob.double() == a.double() @ b.double()).all())


def test_mm_autograd(cuda):
    from sgformer_amd import ops
    g = torch.Generator().manual_seed(5)
    a = torch.randn(37, 21, generator=g).to(cuda).requires_grad_(True)
    b = torch.randn(21, 19, generator=g).to(cuda).requires_grad_(True)
    cot = torch.randn(37, 19, generator=g).to(cuda)
    ga, gb = torch.autograd.grad(ops.mm(a, b), (a, b), cot)
    assert _rel(ga, cot.double() @ b.double().t()) <= 2e-6 and _rel(gb, a.double().t() @ cot.double()) <= 2e-6


# ------------------------------------------------------------------------------------------------
# sgf_pad_rows
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("f", [65, 1433, 3])
@pytest.mark.parametrize("src_dt,dst_dt", [(torch.float32, torch.float32), (torch.float32, torch.bfloat16),
                                           (torch.bfloat16, torch.bfloat16)])
def test_pad_rows(cuda, f, src_dt, dst_dt):
    from sgformer_amd import ops
    g = torch.Generator().manual_seed(f)
    n = 517
    x = torch.randn(n, f, generator=g).to(src_dt).to(cuda)
    fp = (f + 3) // 4 * 4
    out = ops.K.pad_rows(x, None, fp, dst_dt)
    assert out.shape == (n, fp) and out.dtype == dst_dt
    assert bool((out[:, :f] == x.to(dst_dt)).all()) and float(out[:, f:].float().abs().max()) == 0.0
    perm = torch.randperm(n, generator=g).to(torch.int32).to(cuda)
    outp = ops.K.pad_rows(x, perm, fp, dst_dt)
    assert bool((outp[:, :f] == x[perm.long()].to(dst_dt)).all()) and float(outp[:, f:].float().abs().max()) == 0.0
    idx64 = torch.randint(0, n, (100,), generator=g).to(cuda)
    assert bool((ops.K.pad_rows(x, idx64, fp, dst_dt)[:, :f] == x[idx64].to(dst_dt)).all())


# ------------------------------------------------------------------------------------------------
# sgf_attn_h_small_fwd / _bwd
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("d,use_v", [(256, True), (128, True), (64, False), (20, True)])
def test_attention_small_algebra_in_the_library(cuda, d, use_v):
    """The d x d algebra as one library call each way == include/sgf.h's formulas term by term in fp64 (tests/attn_algebra.py)
    — M, m, w, beta and, through autograd on the fp64 form, D = dG + dG^T, ds and the six parameter gradients."""
    from sgformer_amd import ops
    from tests import attn_algebra as A
    g = torch.Generator().manual_seed(d)
    n = 3000
    h = torch.randn(n, d, generator=g, dtype=torch.float64) * 0.5
    G, s = h.t() @ h, h.sum(0)
    par = [torch.randn(d, d, generator=g, dtype=torch.float64) / d ** 0.5 if i % 2 == 0 else
           torch.randn(d, generator=g, dtype=torch.float64) * 0.1 for i in range(6)]
    if not use_v:
        par[4], par[5] = torch.eye(d, dtype=torch.float64), torch.zeros(d, dtype=torch.float64)
    leaves = [t.clone().requires_grad_(True) for t in (G, s, *par)]
    n_total = float(n) * 1.5
    ref = A.attn_h_small(leaves[0], leaves[1], float(n), n_total, *leaves[2:])
    cot = [torch.randn(t.shape, generator=g, dtype=torch.float64) for t in ref]
    gref = torch.autograd.grad(ref, leaves, cot)
    dev = [t.float().to(cuda) for t in (G, s, *par)]
    wv, bv = (dev[6], dev[7]) if use_v else (None, None)
    M, m, w, beta, saved = ops.K.attn_h_small_fwd(dev[0], dev[1], float(n), n_total, dev[2], dev[3], dev[4], dev[5], wv, bv)
    for got, want, name in zip((M, m, w, beta), ref, "M m w beta".split()):
        assert _rel(got.cpu(), want.detach()) <= 2e-5, name
    cd = [t.float().to(cuda) for t in cot]
    Dm, ds, gwq, gbq, gwk, gbk, gwv, gbv = ops.K.attn_h_small_bwd(cd[0].contiguous(), cd[2], cd[1], cd[3], n_total, saved, d, d,
                                                                  want_v=use_v)
    assert _rel(Dm.cpu(), gref[0] + gref[0].t()) <= 2e-4
    assert _rel(ds.cpu(), gref[1]) <= 2e-4
    for got, want, name in zip((gwq, gbq, gwk, gbk), gref[2:6], "gwq gbq gwk gbk".split()):
        assert _rel(got.cpu(), want) <= 2e-4, name
    if use_v:
        assert _rel(gwv.cpu(), gref[6]) <= 2e-4 and _rel(gbv.cpu(), gref[7]) <= 2e-4
    else:
        assert gwv is None and gbv is None
    # run-to-run identical (fixed summation orders everywhere)
    M2, *_ = ops.K.attn_h_small_fwd(dev[0], dev[1], float(n), n_total, dev[2], dev[3], dev[4], dev[5], wv, bv)
    assert bool((M2 == M).all())


# ------------------------------------------------------------------------------------------------
# no library GEMM on any recipe shape
# ------------------------------------------------------------------------------------------------
_LIBRARY = ("Cijk_", "rocblas", "hipblaslt", "Tensile", "gemm_kernel", "cutlass")


def _device_kernels(fn):
    """Names of the device kernels `fn` launches (torch profiler, CUDA = HIP activity)."""
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        fn()
        torch.cuda.synchronize()
    names = [e.name for e in prof.events() if getattr(e, "device_type", None) is not None and "cuda" in str(e.device_type).lower()]
    if not names:
        names = [e.key for e in prof.key_averages() if getattr(e, "device_time_total", 0) or getattr(e, "cuda_time_total", 0)]
    return names


def _assert_no_library_gemm(names):
    assert names, "the profiler listed no device kernels"
    bad = [k for k in names if any(t.lower() in k.lower() for t in _LIBRARY)]
    assert not bad, sorted(set(bad))
    assert any(k.startswith("k_") or "sgf" in k for k in names), sorted(set(names))[:10]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_pokec_recipe_runs_without_a_library_gemm(cuda, dtype):
    """BASELINE config 4's feature width (f = 65, large/run.sh:22-26): x is zero-padded to 68 columns once at the module
    entry and both stems take the aligned kernels; fp32: logits to 1e-4 of the fp64 oracle, gradients relative; the device
    kernel list of forward + backward holds no Tensile / rocBLAS / hipBLASLt kernel."""
    from sgformer_amd import synth
    from sgformer_amd.ours import SGFormer
    cfg = dict(synth.RECIPES["pokec"])
    n, f, d, c = 6000, 65, 256, 2
    torch.manual_seed(2)
    x, ei = torch.randn(n, f), O.synthetic_graph(n, 12.0, seed=3)
    y, idx = torch.randint(0, c, (n,)), torch.randperm(n)[: n // 2]
    p = O.init_params(cfg, f, d, c, seed=5)
    m = SGFormer(f, d, c, trans_dropout=0.0, gnn_dropout=0.0, compute_dtype=None if dtype == torch.float32 else dtype, **cfg)
    m.load_state_dict({**m.state_dict(), **p})
    m = m.to(cuda).train()
    xd, eid, yd, idxd = x.to(cuda), ei.to(cuda), y.to(cuda), idx.to(cuda)
    out = {}

    def step():
        m.zero_grad(set_to_none=True)
        out["logits"] = m(xd, eid)
        O.nll_loss(out["logits"].float(), yd, idxd).backward()

    step()                                      # warm (graph cache, entry copy)
    _assert_no_library_gemm(_device_kernels(step))
    logits = out["logits"]
    p64 = {k: v.double().requires_grad_(v.is_floating_point() and "running" not in k) for k, v in p.items()}
    ref = O.sgformer_forward(p64, x.double(), ei, cfg, training=True)
    O.nll_loss(ref, y, idx).backward()
    err = float((logits.detach().double().cpu() - ref.detach()).abs().max())
    if dtype == torch.float32:
        assert err <= 1e-4, err
        for k, prm in m.named_parameters():
            g = p64[k].grad
            if g is not None and float(g.norm()) > 1e-9:
                assert prm.grad.shape == g.shape
                assert _rel(prm.grad.cpu(), g) <= 1e-3, (k, _rel(prm.grad.cpu(), g))
    else:
        assert err <= 3e-2 * max(1.0, float(ref.detach().abs().max())), err
        for k, prm in m.named_parameters():
            assert prm.grad is not None and prm.grad.shape == prm.shape and bool(torch.isfinite(prm.grad).all()), k


class _Data:
    def __init__(self, x, ei):
        self.graph = {"node_feat": x, "edge_index": ei, "num_nodes": x.shape[0]}


def test_cora_recipe_runs_without_a_library_gemm(cuda):
    """BASELINE config 1 (medium/ours.py, Cora shape: f = 1433, d = 64, C = 7, medium/run.sh:2-7) on the HIP path: the two
    [N, 1433] x [1433, 64] stems run on sgf_gemm, their dW on sgf_gram; fp32 logits within 1e-4 of the fp64 oracle."""
    from sgformer_amd import ours_medium as M
    n, f, d, c = 2708, 1433, 64, 7
    cfg = dict(num_layers=1, alpha=0.5, graph_weight=0.8)
    torch.manual_seed(7)
    gnn = M.GCN(f, d, d, num_layers=4, dropout=0.0)
    m = M.SGFormer(f, d, c, dropout=0.0, gnn=gnn, **cfg)
    x = (torch.rand(n, f) < 0.02).float()
    x = x / x.sum(1, keepdim=True).clamp_min(1.0)
    ei = O.synthetic_graph(n, 3.9, seed=6)[:, :-n]
    y, idx = torch.randint(0, c, (n,)), torch.randperm(n)[:140]
    p = {k: v.detach().clone() for k, v in m.state_dict().items()}
    m = m.to(cuda).train()
    data = _Data(x.to(cuda), ei.to(cuda))
    yd, idxd = y.to(cuda), idx.to(cuda)
    out = {}

    def step():
        m.zero_grad(set_to_none=True)
        out["logits"] = m(data)
        O.nll_loss(out["logits"], yd, idxd).backward()

    step()
    _assert_no_library_gemm(_device_kernels(step))
    p64 = {k: v.double().requires_grad_(v.is_floating_point() and "running" not in k) for k, v in p.items()}
    ref = O.medium_forward(p64, x.double(), ei, cfg, training=True)
    O.nll_loss(ref, y, idx).backward()
    assert float((out["logits"].detach().double().cpu() - ref.detach()).abs().max()) <= 1e-4
    for k, prm in m.named_parameters():
        g = p64[k].grad
        if g is not None and float(g.norm()) > 1e-9:
            assert _rel(prm.grad.cpu(), g) <= 5e-4, (k, _rel(prm.grad.cpu(), g))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_multi_head_and_odd_widths_run_without_a_library_gemm(cuda, dtype):
    """Shapes no recipe uses — two heads (the [d -> 3 H d] projection), hidden width 96, 'cat' aggregate, f = 50 — take the
    general kernel for every Linear; fp32 logits within 1e-4 of the fp64 oracle."""
    from sgformer_amd.ours import SGFormer
    cfg = dict(trans_num_layers=2, trans_num_heads=2, trans_use_bn=True, trans_use_residual=True, trans_use_weight=True,
               trans_use_act=True, gnn_num_layers=2, gnn_use_bn=True, gnn_use_residual=True, gnn_use_weight=True,
               gnn_use_init=True, gnn_use_act=True, use_graph=True, graph_weight=0.6, aggregate="cat")
    n, f, d, c = 1500, 50, 96, 5
    torch.manual_seed(4)
    x, ei = torch.randn(n, f), O.synthetic_graph(n, 7.0, seed=9)
    y, idx = torch.randint(0, c, (n,)), torch.randperm(n)[: n // 2]
    p = O.init_params(cfg, f, d, c, seed=6)
    m = SGFormer(f, d, c, trans_dropout=0.0, gnn_dropout=0.0, compute_dtype=None if dtype == torch.float32 else dtype, **cfg)
    m.load_state_dict({**m.state_dict(), **p})
    m = m.to(cuda).train()
    xd, eid, yd, idxd = x.to(cuda), ei.to(cuda), y.to(cuda), idx.to(cuda)
    out = {}

    def step():
        m.zero_grad(set_to_none=True)
        out["logits"] = m(xd, eid)
        O.nll_loss(out["logits"].float(), yd, idxd).backward()

    step()
    _assert_no_library_gemm(_device_kernels(step))
    p64 = {k: v.double().requires_grad_(v.is_floating_point() and "running" not in k) for k, v in p.items()}
    ref = O.sgformer_forward(p64, x.double(), ei, cfg, training=True)
    O.nll_loss(ref, y, idx).backward()
    err = float((out["logits"].detach().double().cpu() - ref.detach()).abs().max())
    if dtype == torch.bfloat16:
        assert err <= 5e-2 * max(1.0, float(ref.detach().abs().max())), err
        return
    assert err <= 1e-4, err
    for k, prm in m.named_parameters():
        g = p64[k].grad
        if g is not None and float(g.norm()) > 1e-9:
            assert _rel(prm.grad.cpu(), g) <= 1e-3, (k, _rel(prm.grad.cpu(), g))
