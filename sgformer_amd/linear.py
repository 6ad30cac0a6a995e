"""The Linear layers of the path as autograd operators (split out of ops.py in r06): T4 — every y = x W^T + b of
large/ours.py (:123-126, :36-40, :77, :198, :275), the two input stems with their BatchNorm / LayerNorm, and T6 — the GraphConv
layer's Linear + BatchNorm + relu + residual with its chained input gradients.  `ops` re-exports every name."""
from __future__ import annotations

import ctypes
from collections import OrderedDict
from typing import Optional

import torch

from . import _lib
from .kernels import (HipKernels, LONG_ROW, _SEGMENT, _code, _ld, _one_pass_cat, _pair_gram, _ptr, _rows,  # noqa: F401
                      _rows16, _stream, _workspace, _workspaces)

_F32 = torch.float32
_BF16 = torch.bfloat16

from .graph import K  # noqa: E402,F401  (ONE kernel table, rebound in every module by ops.set_kernels)


# BatchNorm's batch statistics of a tensor (large/ours.py:77-81, :87-88): shifted one-pass column sums
_BN_SAMPLE_ROWS = 1024


def batch_stats(x: torch.Tensor, shard=None):
    """Column mean / biased variance over all rows (all ranks when sharded) in ONE pass over x.

    Shifted single pass: the shift is the column mean of a small leading sample of rows (a ~10 us
    kernel), then one streaming pass accumulates [sum (x - s) | sum (x - s)^2] in fp32;
    mean = s + m1, var = m2 - m1^2 with m1 ~ sigma / sqrt(sample) — no cancellation, unlike the
    unshifted E[x^2] - E[x]^2, and half the HBM traffic of the two-pass form.  Node-sharded runs
    all-reduce the sample sums (so that every rank uses the same shift) and then the two sums."""
    K.check(x)
    x = _rows(x.detach())
    n, d = x.shape
    ns = min(n, _BN_SAMPLE_ROWS)
    samp = torch.cat([K.colstats(x[:ns], None)[:d], torch.full((1,), float(ns), dtype=_F32, device=x.device)])
    n_tot = float(n)
    if shard is not None:
        shard.all_reduce(samp)
        n_tot = float(shard.n_global)
    shift = (samp[:d] / samp[d].clamp_min(1.0)).contiguous()
    st = K.colstats(x, shift)
    if shard is not None:
        shard.all_reduce(st)
    m1 = st[:d] / max(n_tot, 1.0)
    mean = shift + m1
    var = (st[d:] / max(n_tot, 1.0) - m1 * m1).clamp_min_(0.0)
    return mean, var, n_tot


# ------------------------------------------------------------------------------------------------
# T4: the Linear layers  y = x W^T + b  (large/ours.py:123-126, :36-40, :77, :198, :275).
# bf16 storage, square layers of width 64 / 128 / 256 (and pairs [x1 | x2] of them): forward, the BatchNorm column sums of
# the output and dX on the streaming row kernels (sgf_gcn_epilogue_*, csrc/rowgemm.hip); fp32 storage, widths % 4 up to 256:
# the exact-fp32 streaming kernel (csrc/linear_f32.hip); EVERY other shape: the general matrix-core kernel sgf_gemm
# (csrc/gemm.hip) — no Linear of the path is a library GEMM.  The weight / bias gradients
#     dW = dY^T X   (a d x d <- [N x d]^T [N x d] reduction over all nodes),   db = colsum(dY)
# always run on sgf_gram: hipBLASLt's kernels for that shape ran at 0.7 TB/s (3.6 ms per call at
# ogbn-products scale, profiles/r01_products_bf16_kernel_stats.md) and ATen's column reduction for
# the bias gradient at 18.6 ms for C = 47.  Master weights may be fp32 while activations are bf16.
# ------------------------------------------------------------------------------------------------
class _Linear(torch.autograd.Function):
    """y = sum_i x_i W_i^T + b with W = [W_1 | W_2 | ...] split along its input dimension
    (one operand for nn.Linear; two for GraphConvLayer's W [A x | x0], large/ours.py:36-38,
    without materialising the concatenation)."""

    @staticmethod
    def forward(ctx, w, b, stats_req, *xs):
        K.check(*xs)
        dt = xs[0].dtype
        wc = w if w.dtype == dt else w.to(dt)
        widths = [x.shape[1] for x in xs]
        if sum(widths) != w.shape[1]:
            raise RuntimeError(f"linear: input widths {widths} do not add up to {w.shape[1]}")
        offs = [sum(widths[:i]) for i in range(len(widths))]
        fused = (len(xs) <= 2 and all(_streaming_linear_ok(x, wc[:, o:o + k]) for x, o, k in zip(xs, offs, widths)))
        if fused:
            # streaming passes with W resident in LDS (sgf_gcn_epilogue_*); the BatchNorm that follows gets its
            # column sums from the same pass
            xr = [_rows16(x) for x in xs]
            b32 = None if b is None else b.detach().float().contiguous()
            if stats_req is not None:
                y = _linear_with_stats(xr, wc, b32, stats_req)
            else:
                y, _ = _streaming_linear(xr, wc, b32)
        else:
            # any other shape (input widths beyond 256 or not multiples of 4, odd hidden widths, multi-head projections):
            # the general matrix-core kernel (sgf_gemm, csrc/gemm.hip) — first operand with the bias, the rest accumulated
            # IN PLACE; W's column blocks are passed as strided views (no copy, no transposition)
            b32 = None if b is None else b.detach().float().contiguous()
            y = K.gemm(xs[0], wc[:, :widths[0]].t(), bias=b32)
            off = widths[0]
            for x, k in zip(xs[1:], widths[1:]):
                K.gemm(x, wc[:, off:off + k].t(), out=y, beta=1.0, addend=y)
                off += k
        if stats_req is not None and not fused:
            stats_req["out"] = batch_stats(y, stats_req.get("shard"))
        ctx.save_for_backward(wc, *xs)
        ctx.meta = (w.dtype, None if b is None else b.dtype, widths)
        return y

    @staticmethod
    def backward(ctx, g):
        wc, *xs = ctx.saved_tensors
        wdtype, bdtype, widths = ctx.meta
        g = g.contiguous()
        need_w, need_b = ctx.needs_input_grad[0], ctx.needs_input_grad[1] and bdtype is not None
        dxs, off = [], 0
        for i, k in enumerate(widths):
            if not ctx.needs_input_grad[3 + i]:
                dxs.append(None)
            elif _streaming_linear_ok(g, wc[:, off:off + k], dx=True):
                dxs.append(K.gcn_epilogue_dx(_rows16(g), wc[:, off:off + k]))
            else:
                dxs.append(K.gemm(g, wc[:, off:off + k]))
            off += k
        dw, db = _linear_param_grads(g, xs, widths, need_w, need_b, wdtype, bdtype)
        return (dw, db, None, *dxs)


def _linear_param_grads(g, xs, widths, need_w, need_b, wdtype, bdtype):
    """dW = g^T [x_1 | x_2 | ...], db = sum_n g on sgf_gram (one node reduction per operand)."""
    dw = db = None
    if need_w or need_b:
        # sgf_gram wants widths that are multiples of 4 elements: zero-pad the odd ones (e.g. the
        # C = 47 logits gradient: one extra [N, 48] pass) and slice the result
        m = g.shape[1]
        gp = _rows(g if m % 4 == 0 else torch.nn.functional.pad(g, (0, 4 - m % 4)))
        dw = torch.empty((gp.shape[1], sum(widths)), dtype=_F32, device=g.device)
        off = 0
        if (len(xs) == 2 and widths[0] == widths[1] and widths[0] % 4 == 0 and hasattr(K, "gram2") and gp.dtype == _BF16
                and xs[0].dtype == _BF16 and xs[1].dtype == _BF16 and _pair_gram()):
            # both blocks of dW = g^T [x_1 | x_2] from ONE read of g out of HBM (paired launch, sgf_gram2)
            k = widths[0]
            db = K.gram2(gp, _rows(xs[0]), _rows(xs[1]), dw[:, :k], dw[:, k:], want_colsum=need_b)
            xs, widths = [], []
        for i, (x, k) in enumerate(zip(xs, widths)):
            if k % 4 == 0 and off % 4 == 0:   # sgf_gram stores float4s: the slice must stay 16-B aligned
                _, cs = K.gram(gp, _rows(x), out=dw[:, off:off + k], want_colsum=(i == 0 and need_b))
            elif k % 4 == 0:
                blk, cs = K.gram(gp, _rows(x), want_colsum=(i == 0 and need_b))
                dw[:, off:off + k] = blk
            else:
                xp = _rows(torch.nn.functional.pad(x, (0, 4 - k % 4)))
                blk, cs = K.gram(gp, xp, want_colsum=(i == 0 and need_b))
                dw[:, off:off + k] = blk[:, :k]
            if i == 0:
                db = cs
            off += k
        dw = dw[:m].to(wdtype) if need_w else None
        db = db[:m].to(bdtype) if need_b else None
    return dw, db


class _StemPair(torch.autograd.Function):
    """(y0, y1) = (x W0^T + b0, x W1^T + b1) from ONE pass over x — the first Linear of GraphConv and of TransConv
    (large/ours.py:77, :198) read the same node features; y0's BatchNorm column sums ride along (stats_req)."""

    @staticmethod
    def forward(ctx, x, w0, b0, w1, b1, stats_req):
        K.check(x)
        dt = x.dtype
        w0c, w1c = w0.to(dt), w1.to(dt)
        f32 = [None if b is None else b.detach().float().contiguous() for b in (b0, b1)]
        d = w0.shape[0]
        shift = None
        if stats_req is not None:
            shard = stats_req.get("shard")
            n = x.shape[0]
            ns = min(n, _BN_SAMPLE_ROWS)
            _, _, st_s = K.stem_pair(x[:ns], w0c, f32[0], None, None, None, want_stats0=True)
            samp = torch.cat([st_s[:d], torch.full((1,), float(ns), dtype=_F32, device=x.device)])
            n_tot = float(n)
            if shard is not None:
                shard.all_reduce(samp)
                n_tot = float(shard.n_global)
            shift = (samp[:d] / samp[d].clamp_min(1.0)).contiguous()
        y0, y1, st = K.stem_pair(x, w0c, f32[0], w1c, f32[1], shift, want_stats0=stats_req is not None)
        if stats_req is not None:
            if shard is not None:
                shard.all_reduce(st)
            m1 = st[:d] / max(n_tot, 1.0)
            stats_req["out"] = (shift + m1, (st[d:] / max(n_tot, 1.0) - m1 * m1).clamp_min_(0.0), n_tot)
        ctx.save_for_backward(x, w0c, w1c)
        ctx.meta = (w0.dtype, None if b0 is None else b0.dtype, w1.dtype, None if b1 is None else b1.dtype)
        return y0, y1

    @staticmethod
    def backward(ctx, g0, g1):
        x, w0c, w1c = ctx.saved_tensors
        wd0, bd0, wd1, bd1 = ctx.meta
        k = [x.shape[1]]
        dw0, db0 = _linear_param_grads(g0.contiguous(), [x], k, ctx.needs_input_grad[1],
                                       ctx.needs_input_grad[2] and bd0 is not None, wd0, bd0)
        dw1, db1 = _linear_param_grads(g1.contiguous(), [x], k, ctx.needs_input_grad[3],
                                       ctx.needs_input_grad[4] and bd1 is not None, wd1, bd1)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = K.gemm(g0.contiguous(), w0c)
            K.gemm(g1.contiguous(), w1c, out=dx, beta=1.0, addend=dx)
        return dx, dw0, db0, dw1, db1, None


class _StemPairBN(torch.autograd.Function):
    """(x0, x0, y1): x0 = relu(BatchNorm(x W0^T + b0)) — GraphConv's stem, large/ours.py:77-80 — handed out TWICE (the first
    SpMM and the layers' Linear / residual consume it; their gradients come back separately instead of through an add),
    y1 = x W1^T + b1 (TransConv's stem, :198) from the same read of x.  Backward: the BatchNorm's two sums over both
    gradients (sgf_bn_bwd_stats2), then dW0 / db0 straight from them — dz is formed inside the Gram kernel and never written
    (sgf_gram_bn_bwd): x is data, nobody else needs dz."""

    @staticmethod
    def forward(ctx, x, w0, b0, w1, b1, gamma, beta, bn_hook, shard, ln_gamma=None, ln_beta=None, ln_cfg=None):
        """ln_cfg = (eps, relu, affine) — not None: the third output is [relu](LayerNorm(y1)) (TransConv's stem, large/ours.py:
        198-201) instead of y1, and its backward takes dW1 / db1 / d ln_gamma / d ln_beta from sgf_gram_ln_bwd."""
        K.check(x)
        dt = x.dtype
        w0c, w1c = w0.to(dt), w1.to(dt)
        f32 = [None if b is None else b.detach().float().contiguous() for b in (b0, b1)]
        d = w0.shape[0]
        n = x.shape[0]
        want = bn_hook(None)
        shift = None
        if want:
            ns = min(n, _BN_SAMPLE_ROWS)
            _, _, st_s = K.stem_pair(x[:ns], w0c, f32[0], None, None, None, want_stats0=True)
            n_tot = float(n)
            if shard is None:
                shift = st_s[:d] * (1.0 / float(max(ns, 1)))
            else:
                samp = torch.cat([st_s[:d], torch.full((1,), float(ns), dtype=_F32, device=x.device)])
                shard.all_reduce(samp)
                n_tot = float(shard.n_global)
                shift = (samp[:d] / samp[d].clamp_min(1.0)).contiguous()
        y0, y1, st = K.stem_pair(x, w0c, f32[0], w1c, f32[1], shift, want_stats0=want)
        if want:
            if shard is not None:
                shard.all_reduce(st)
            if hasattr(K, "bn_finalize"):
                mean, rstd, n_tot, training = bn_hook(("raw", st, shift, n_tot))
            else:
                m1 = st[:d] / max(n_tot, 1.0)
                mean, rstd, n_tot, training = bn_hook((shift + m1, (st[d:] / max(n_tot, 1.0) - m1 * m1).clamp_min_(0.0), n_tot))
        else:
            mean, rstd, n_tot, training = bn_hook(False)
        g32 = gamma.detach().float().contiguous() if gamma is not None else None
        be32 = beta.detach().float().contiguous() if beta is not None else None
        mean = mean.detach().float().contiguous()
        rstd = rstd.detach().float().contiguous()
        x0 = K.bn_apply(y0, mean, rstd, g32, be32, None, True)
        out1, ln_saved = y1, (None, None, None, None, None)
        if ln_cfg is not None:
            eps, ln_relu, affine = ln_cfg
            lg32 = ln_gamma.detach().float().contiguous() if affine else None
            lb32 = ln_beta.detach().float().contiguous() if affine else None
            out1, lmean, lrstd = K.ln_fwd(_rows(y1), None, 1.0, 0.0, lg32, lb32, bool(ln_relu), float(eps))
            ln_saved = (y1, lmean, lrstd, lg32, lb32)
        ctx.save_for_backward(x, w0c, w1c, y0, g32, be32, mean, rstd, *ln_saved)
        ctx.meta = (w0.dtype, None if b0 is None else b0.dtype, w1.dtype, None if b1 is None else b1.dtype,
                    None if gamma is None else gamma.dtype, bool(training), float(n_tot), shard,
                    None if ln_cfg is None else (bool(ln_cfg[1]), None if ln_gamma is None else ln_gamma.dtype))
        return x0, x0.view_as(x0), out1

    @staticmethod
    def backward(ctx, ga, gb, g1):
        x, w0c, w1c, y0, g32, be32, mean, rstd, y1, lmean, lrstd, lg32, lb32 = ctx.saved_tensors
        wd0, bd0, wd1, bd1, gdt, training, n_tot, shard, ln_meta = ctx.meta
        d = y0.shape[1]
        if ga is None:
            ga, gb = gb, None
        if ga is None:
            ga = torch.zeros_like(y0)
        ga = _rows(ga.contiguous())
        gb = None if gb is None else _rows(gb.contiguous())
        stats = K.bn_bwd_stats2(ga, gb, y0, mean, rstd, g32, be32, True)
        if shard is not None:
            shard.all_reduce(stats)
        inv_n = 1.0 / max(n_tot, 1.0)
        dw0, db0 = K.gram_bn_bwd(ga, gb, y0, mean, rstd, g32, be32, True, stats, inv_n, training, _rows(x))
        dw0 = dw0.to(wd0) if ctx.needs_input_grad[1] else None
        db0 = db0.to(bd0) if (ctx.needs_input_grad[2] and bd0 is not None) else None
        dlg = dlb = None
        if ln_meta is not None:
            # TransConv's stem: LayerNorm backward inside the Gram (dl never written); row-local, no collective
            ln_relu, lgdt = ln_meta
            dw1, db1, dlg, dlb = K.gram_ln_bwd(_rows(g1.contiguous()), y1, lmean, lrstd, lg32, lb32, ln_relu, _rows(x))
            dw1 = dw1.to(wd1) if ctx.needs_input_grad[3] else None
            db1 = db1.to(bd1) if (ctx.needs_input_grad[4] and bd1 is not None) else None
            dlg, dlb = (dlg.to(lgdt), dlb.to(lgdt)) if lg32 is not None else (None, None)
        else:
            dw1, db1 = _linear_param_grads(g1.contiguous(), [x], [x.shape[1]], ctx.needs_input_grad[3],
                                           ctx.needs_input_grad[4] and bd1 is not None, wd1, bd1)
        dgamma = stats[d:].to(gdt) if g32 is not None else None
        dbeta = stats[:d].to(gdt) if be32 is not None else None
        if shard is not None and g32 is not None:
            dgamma, dbeta = shard.unsum(dgamma), shard.unsum(dbeta)
        dx = None
        if ctx.needs_input_grad[0]:                      # features that require a gradient (not in any recipe): explicit dz
            g = ga if gb is None else ga + gb
            dz = K.bn_bwd_apply(g, y0, mean, rstd, g32, be32, True, stats, inv_n, training)
            gl = g1
            if ln_meta is not None:
                gl = K.ln_bwd(_rows(g1.contiguous()), None, y1, None, 1.0, 0.0, lg32, False, lmean, lrstd)[0] \
                    if not ln_meta[0] else None
                if gl is None:
                    raise NotImplementedError("features that require a gradient behind the fused LayerNorm stem: set "
                                              "SGF_STEM_LN_FUSED=0")
            dx = K.gemm(dz, w0c)
            K.gemm(gl.contiguous(), w1c, out=dx, beta=1.0, addend=dx)
        return dx, dw0, db0, dw1, db1, dgamma, dbeta, None, None, dlg, dlb, None


def stem_pair_bn_supported(x, w0, w1) -> bool:
    return (stem_pair_supported(x, w0, w1) and hasattr(K, "gram_bn_bwd_supported")
            and K.gram_bn_bwd_supported(w0.shape[0], x.shape[1], x.dtype))


def stem_pair_bn(x, w0, b0, w1, b1, gamma, beta, bn_hook, shard=None, ln=None):
    """(x0 for the layers, x0 for the first SpMM, y1): see _StemPairBN.  ln = (LayerNorm weight, bias, eps, relu): the third
    output is [relu](LayerNorm(y1)) — TransConv's stem — when the shapes allow (stem_ln_supported)."""
    if ln is None:
        return _StemPairBN.apply(x, w0, b0, w1, b1, gamma, beta, bn_hook, shard)
    lg, lb, eps, relu = ln
    return _StemPairBN.apply(x, w0, b0, w1, b1, gamma, beta, bn_hook, shard, lg, lb, (eps, relu, lg is not None))


def stem_ln_supported(x, w1) -> bool:
    import os
    return (hasattr(K, "gram_ln_bwd_supported") and K.gram_ln_bwd_supported(w1.shape[0], x.shape[1], x.dtype)
            and not x.requires_grad and os.environ.get("SGF_STEM_LN_FUSED", "1") != "0")


def stem_pair_supported(x, w0, w1) -> bool:
    return (x.dim() == 2 and x.shape[0] > 0 and w0.shape == w1.shape and x.stride(-1) == 1
            and (x.stride(0) * x.element_size()) % 8 == 0 and x.data_ptr() % 8 == 0
            and K.stem_pair_supported(x.shape[1], w0.shape[0], x.dtype))


def stem_pair(x, w0, b0, w1, b1, want_stats0=False, shard=None):
    """((y0, y1), stats0): both input stems from one read of x; stats0 = (mean, var, count) of y0 for its BatchNorm
    when asked for, else None."""
    req = {"shard": shard, "out": None} if want_stats0 else None
    y0, y1 = _StemPair.apply(x, w0, b0, w1, b1, req)
    return (y0, y1), (req["out"] if req is not None else None)


def _streaming_linear_ok(x: torch.Tensor, wc: torch.Tensor, dx: bool = False) -> bool:
    """x wc^T (or, dx=True, x wc) on the streaming kernels: bf16 square layers, fp32 layers of widths % 4 up to 256."""
    return (x.dim() == 2 and x.shape[0] > 0 and wc.stride(-1) == 1 and (wc.stride(0) * wc.element_size()) % 16 == 0
            and wc.data_ptr() % 16 == 0 and x.shape[1] == (wc.shape[0] if dx else wc.shape[1])
            and K.gcn_epilogue_supported(wc.shape[1], wc.shape[0], x.dtype))


def _streaming_linear(xr, wc, b32, shift=None, want_stats=False, rows=None):
    """[x_1 | x_2] wc^T + b32 on the first `rows` rows (all by default): one or two streaming passes."""
    xs = xr if rows is None else [x[:rows] for x in xr]
    if len(xs) == 1:
        return K.gcn_epilogue_stats(xs[0], wc, b32, shift, want_stats=want_stats)
    return K.gcn_epilogue_cat(xs[0], xs[1], wc, b32, shift, want_stats=want_stats)


def _linear_with_stats(xr, wc, b32, stats_req):
    """y = [x_1 | x_2] wc^T + b32 AND BatchNorm's batch statistics of y, from the same pass.  Same shifted sums as
    batch_stats: the shift is the column mean of the first rows of y, which a small launch over those rows provides."""
    shard = stats_req.get("shard")
    n, d = xr[0].shape[0], wc.shape[0]
    ns = min(n, _BN_SAMPLE_ROWS)
    _, st_s = _streaming_linear(xr, wc, b32, None, want_stats=True, rows=ns)
    n_tot = float(n)
    if shard is None:
        shift = st_s[:d] * (1.0 / float(max(ns, 1)))          # one launch; the sharded form needs the global sample count
    else:
        samp = torch.cat([st_s[:d], torch.full((1,), float(ns), dtype=_F32, device=xr[0].device)])
        shard.all_reduce(samp)
        n_tot = float(shard.n_global)
        shift = (samp[:d] / samp[d].clamp_min(1.0)).contiguous()
    y, st = _streaming_linear(xr, wc, b32, shift, want_stats=True)
    if shard is not None:
        shard.all_reduce(st)
    if stats_req.get("raw"):
        stats_req["out"] = ("raw", st, shift, n_tot)              # the caller finalises (K.bn_finalize: one launch)
        return y
    m1 = st[:d] / max(n_tot, 1.0)
    stats_req["out"] = (shift + m1, (st[d:] / max(n_tot, 1.0) - m1 * m1).clamp_min_(0.0), n_tot)
    return y


# ------------------------------------------------------------------------------------------------
# T4 + T6 as ONE autograd node: out = [relu](BatchNorm(W [y | x0] + b)) [+ x0]   (large/ours.py:36-40, 87-93)
# ------------------------------------------------------------------------------------------------
class GradChain:
    """The gradient of x0 = layer_[0] of one GraphConv forward, collected across its layers' backward nodes (which run in
    reverse layer order: layer i's input is layer i-1's output) and returned to autograd ONCE, by the first layer's node —
    the last to run; the others return nothing for x0.  Two forms:
      * fused kernel (SGF_GCN_BWD_FUSED=1): every layer's sgf_gcn_bn_bwd_dx adds its two contributions (the residual's gy
        and dz W[:, d:]) to the running sum it is handed (`acc`, opaque);
      * default: the contributions are kept (`parts`) and summed in one pass (sgf_sum_n) by the first layer's node."""

    def __init__(self):
        self.acc = None
        self.parts = []


def _fused_bwd() -> bool:
    """sgf_gcn_bn_bwd_dx in the layers' backward.  Off by default: at d = 256 the four workgroups per row tile do not stay
    inside the L2's window on their own (gy / z leave HBM four times) and with the per-tile rendezvous the launch is bound
    by its own serial phases — 2.4-4.4 ms against 2.1 ms for the separate kernels (profiles/r04_bn_bwd_dx_*.md)."""
    import os
    return os.environ.get("SGF_GCN_BWD_FUSED", "0") == "1"


def _acc_in_place(d: int, dtype) -> bool:
    """sgf_gcn_epilogue_dx2_acc in the layers' backward (default on; SGF_GCN_DX_ACC=0: paired dx2 + one sgf_sum_n)."""
    import os
    return (hasattr(K, "gcn_epilogue_dx2_acc_supported") and K.gcn_epilogue_dx2_acc_supported(d, dtype)
            and os.environ.get("SGF_GCN_DX_ACC", "1") != "0")


def gcn_layer_fused_ok(x0: torch.Tensor, w: torch.Tensor) -> bool:
    """bf16 storage, square blocks of 64 / 128 / 256, W = [W1 | W2] — what sgf_gcn_bn_bwd_dx / sgf_gcn_epilogue_cat take."""
    import os
    d = x0.shape[1] if x0.dim() == 2 else 0
    return (x0.dim() == 2 and x0.shape[0] > 0 and x0.dtype == _BF16 and tuple(w.shape) == (d, 2 * d)
            and hasattr(K, "gcn_bn_bwd_dx_supported") and K.gcn_bn_bwd_dx_supported(d, x0.dtype)
            and os.environ.get("SGF_GCN_FUSED", "1") != "0")


class _LinearBNActRes(torch.autograd.Function):
    """y = A x (already multiplied), x0 -> z = [y | x0] W^T + b -> out = [relu](BatchNorm(z)) [+ x0].

    Forward: one pass for the Linear and BatchNorm's batch sums (sgf_gcn_epilogue_cat), `bn_hook(stats)` — the module's
    own bookkeeping (batch vs running statistics, running-stat update) — then sgf_bn_apply.
    Backward: sgf_bn_bwd_stats (the one global reduction), then ONE launch for dz, d y and the running gradient of x0
    (sgf_gcn_bn_bwd_dx), then the weight / bias gradients on sgf_gram."""

    @staticmethod
    def forward(ctx, y, x0, w, b, gamma, beta, bn_hook, relu: bool, use_res: bool, shard, chain, first: bool):
        K.check(y, x0)
        dt = x0.dtype
        wc = w.detach().to(dt).contiguous()
        b32 = None if b is None else b.detach().float().contiguous()
        yr, xr = _rows16(y), _rows16(x0)
        want = bn_hook(None)                         # does the BatchNorm normalise with batch statistics?
        if want:
            req = {"shard": shard, "out": None, "raw": hasattr(K, "bn_finalize")}
            z = _linear_with_stats([yr, xr], wc, b32, req)
            mean, rstd, n_tot, training = bn_hook(req["out"])
        else:
            z, _ = _streaming_linear([yr, xr], wc, b32)
            mean, rstd, n_tot, training = bn_hook(False)
        g32 = gamma.detach().float().contiguous() if gamma is not None else None
        be32 = beta.detach().float().contiguous() if beta is not None else None
        mean = mean.detach().float().contiguous()
        rstd = rstd.detach().float().contiguous()
        out = K.bn_apply(z, mean, rstd, g32, be32, xr if use_res else None, relu)
        ctx.save_for_backward(yr, xr, z, wc, g32, be32, mean, rstd)
        ctx.meta = (bool(relu), bool(use_res), bool(training), float(n_tot), shard, chain, bool(first), w.dtype,
                    None if b is None else b.dtype, None if gamma is None else gamma.dtype)
        return out

    @staticmethod
    def backward(ctx, gout):
        yr, xr, z, wc, g32, be32, mean, rstd = ctx.saved_tensors
        relu, use_res, training, n_tot, shard, chain, first, wdt, bdt, gdt = ctx.meta
        d = z.shape[1]
        gout = _rows16(gout.contiguous())
        stats = K.bn_bwd_stats(gout, z, mean, rstd, g32, be32, relu)
        if shard is not None:
            shard.all_reduce(stats)
        inv_n = 1.0 / max(n_tot, 1.0)
        if _fused_bwd():
            dz, dy, acc = K.gcn_bn_bwd_dx(gout, z, mean, rstd, g32, be32, relu, stats, inv_n, training, wc, chain.acc,
                                          last=first, add_gy=use_res)
            chain.acc = None if first else acc
            dx0 = acc if first else None
        else:
            # BatchNorm backward, then BOTH input gradients from one HBM read of dz (paired launch); x0's contributions wait
            # in the chain for the one summation pass
            fused_dw = None
            if (ctx.needs_input_grad[2] and hasattr(K, "gram2_bn_bwd_supported") and _pair_gram()
                    and K.gram2_bn_bwd_supported(gout, z, yr, xr)):
                # dz, dW = dz^T [y | x0] and db from ONE pass over (gout, z): sgf_gram2_bn_bwd (csrc/gramx.hip, k_gramb2)
                dwf = torch.empty((d, 2 * d), dtype=_F32, device=z.device)
                dz, dbf = K.gram2_bn_bwd(gout, z, mean, rstd, g32, be32, relu, stats, inv_n, training, yr, xr, dwf[:, :d], dwf[:, d:])
                fused_dw = (dwf.to(wdt), dbf.to(bdt) if (ctx.needs_input_grad[3] and bdt is not None) else None)
            else:
                dz = K.bn_bwd_apply(gout, z, mean, rstd, g32, be32, relu, stats, inv_n, training)
            if _acc_in_place(d, dz.dtype) and not chain.parts:
                # x0's gradient accumulated IN PLACE: this layer's dz W2 and residual gradient join the running sum inside
                # the launch that produces dy (balanced pair of workgroups at d = 256) — no k-operand summation pass at the end
                dy, chain.acc = K.gcn_epilogue_dx2_acc(dz, wc, _rows16(gout) if use_res else None, chain.acc)
                dx0 = None
                if first:
                    dx0, chain.acc = chain.acc, None
            else:
                dy, dxi = K.gcn_epilogue_dx2(dz, wc[:, :d], wc[:, d:], True)
                chain.parts.append(dxi)
                if use_res:
                    chain.parts.append(gout)
                dx0 = None
                if first:
                    parts, chain.parts = chain.parts, []
                    if chain.acc is not None:
                        parts.append(chain.acc)
                        chain.acc = None
                    dx0 = parts[0] if len(parts) == 1 else (K.sum_n(parts) if len(parts) <= 8 else sum(parts[1:], parts[0]))
        if not _fused_bwd() and fused_dw is not None:
            dw, db = fused_dw
        else:
            dw, db = _linear_param_grads(dz, [yr, xr], [d, d], ctx.needs_input_grad[2],
                                         ctx.needs_input_grad[3] and bdt is not None, wdt, bdt)
        dgamma = stats[d:].to(gdt) if g32 is not None else None
        dbeta = stats[:d].to(gdt) if be32 is not None else None
        if shard is not None and g32 is not None:
            dgamma, dbeta = shard.unsum(dgamma), shard.unsum(dbeta)
        return dy, dx0, dw, db, dgamma, dbeta, None, None, None, None, None, None


def linear_bn_act_res(y, x0, w, b, gamma, beta, bn_hook, relu, use_res, shard, chain, first):
    return _LinearBNActRes.apply(y, x0, w, b, gamma, beta, bn_hook, relu, use_res, shard, chain, first)


def linear(x, w, b):
    """nn.Linear: square bf16 layers / fp32 layers up to 256 wide on the streaming row kernels (sgf_gcn_epilogue_stats / _dx),
    every other shape on sgf_gemm; weight / bias gradients on sgf_gram."""
    return _Linear.apply(w, b, None, x)


def linear_bn_stats(xs, w, b, shard=None):
    """(y, (mean, var, n_tot)): nn.Linear of x (or of [x_1 | x_2] for a tuple, GraphConvLayer's use_init) and the
    batch statistics BatchNorm1d needs of its output (large/ours.py:36-40 followed by :87-88) — from the Linear's own
    pass when its blocks are square bf16, else linear + batch_stats."""
    req = {"shard": shard, "out": None}
    xs = xs if isinstance(xs, (tuple, list)) else (xs,)
    y = _Linear.apply(w, b, req, *xs)
    return y, req["out"]


def linear_cat(xs, w, b):
    """[x_1 | x_2 | ...] W^T + b without the concatenation (GraphConvLayer with use_init)."""
    return _Linear.apply(w, b, None, *xs)


def out_linear_cat(xs, w, b):
    """The output head of aggregate='cat' (large/ours.py:271-275: fc(cat(x1, x2))) WITHOUT the [N, 2 d] concatenation:
    W = [W_1 | W_2] applied operand by operand (fp32 storage: two passes on the exact-fp32 matrix cores, the first product
    parked as an [N, C] fp32 partial — small next to the [N, d] operands; bf16: GEMM + GEMM(beta = 1)).  Class counts that
    are not multiples of 4 are padded with zero rows for the fp32 kernel, as in out_linear."""
    m = w.shape[0]
    if (xs[0].dtype == _F32 and xs[0].is_cuda and m % 4 != 0
            and all(K.gcn_epilogue_supported(x.shape[1], (m + 3) // 4 * 4, _F32) for x in xs)):
        pad = (m + 3) // 4 * 4 - m
        wp = torch.nn.functional.pad(w, (0, 0, 0, pad))
        bp = None if b is None else torch.nn.functional.pad(b, (0, pad))
        return _Linear.apply(wp, bp, None, *xs)[:, :m]
    return _Linear.apply(w, b, None, *xs)


def out_linear(x, w, b):
    """The output head (large/ours.py:275).  fp32 storage and a class count that is not a multiple of 4 (C = 47): W and b
    are padded with zero rows to the next multiple and the result sliced, so that the layer still runs on the streaming
    fp32 kernel (csrc/linear_f32.hip) instead of a library GEMM; autograd slices the gradients back."""
    m = w.shape[0]
    if (x.dtype == _F32 and x.is_cuda and x.dim() == 2 and m % 4 != 0
            and K.gcn_epilogue_supported(w.shape[1], (m + 3) // 4 * 4, _F32)):
        pad = (m + 3) // 4 * 4 - m
        wp = torch.nn.functional.pad(w, (0, 0, 0, pad))
        bp = None if b is None else torch.nn.functional.pad(b, (0, pad))
        return _Linear.apply(wp, bp, None, x)[:, :m]
    if x.dtype == _BF16 and x.dim() == 2 and m % 8 != 0 and not K.gcn_epilogue_supported(w.shape[1], m, _BF16):
        # bf16 storage and a class count whose rows would not be 16-byte aligned (the 100M recipe: C = 172 -> 344-byte rows):
        # W / b padded with zero rows to the next multiple of 8, so that sgf_gemm stages the logits' gradient with 16-byte
        # loads in the backward (dx = g W) as it stages x in the forward; the result is sliced, autograd slices back
        pad = (m + 7) // 8 * 8 - m
        wp = torch.nn.functional.pad(w, (0, 0, 0, pad))
        bp = None if b is None else torch.nn.functional.pad(b, (0, pad))
        return _Linear.apply(wp, bp, None, x)[:, :m]
    return _Linear.apply(w, b, None, x)
