"""Autograd-level operators over the libsgf C ABI (include/sgf.h).

Each operator is the reference arithmetic of one row of SURVEY.md §8a, forward and backward, as a
`torch.autograd.Function`.  The Functions own the autograd bookkeeping, the multi-GPU exchange
points and nothing else; every tensor-to-tensor computation goes through the kernel table `K`
(class `HipKernels`), whose methods are thin ctypes calls into libsgf.so on the current stream.
PyTorch is used for memory, streams and autograd only; there is no eager fallback — a CPU tensor
or a missing library raises.

(The kernel table is a seam for tests: tests/ may install a CPU table built on oracle/ with
`set_kernels()` to exercise this file, ours.py and dist.py under gloo without a GPU.  No such
table ships in the package.)

Node-sharded multi-GPU runs (sgformer_amd/dist.py) pass a `ShardContext`; the operators then
all-reduce exactly the partial-sum buffers the kernels were designed around (attention stats,
BatchNorm statistics) and all-gather the SpMM operand.
"""
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


from . import graph as _graph
from .graph import *  # noqa: F401,F403,E402
from .graph import K, graph_cache, _own_block_spmm, _sharded_spmm, _tile_params, _block_shape, _mode  # noqa: F401,E402


def set_kernels(table):
    """Install a kernel table (tests only; see module docstring).  Returns the previous one."""
    global K
    prev, K = K, table
    _graph.K = _linear.K = table             # ONE table: graph.py and linear.py look theirs up by the same name
    graph_cache.clear()
    return prev


def _require_cuda(*tensors):
    K.check(*tensors)


# ------------------------------------------------------------------------------------------------
# T3: linear global attention (large/ours.py:130-157)
# ------------------------------------------------------------------------------------------------
def _split(qkv, v_ext, heads, d):
    hd = heads * d
    q, k = qkv[:, :hd], qkv[:, hd:2 * hd]
    if v_ext is None:
        return q, k, qkv[:, 2 * hd:], heads
    return q, k, v_ext, 1


class _Attention(torch.autograd.Function):
    """qkv: [N, 3*H*d] = [Q | K | V], or [N, 2*H*d] = [Q | K] with v_ext: [N, d] (V not projected).
    per_head (H > 1): returns the per-head outputs [N, H*d] (what full_attention_conv returns, medium/ours.py:14-46) and
    takes their gradient [N, H*d] in the backward (sgf_attn_bwd_*_heads) — instead of the head mean [N, d]."""

    @staticmethod
    def forward(ctx, qkv, v_ext, heads: int, d: int, shard, n_override=None, per_head=False):
        K.check(qkv, v_ext)
        qkv, v_ext = _rows(qkv), _rows(v_ext)
        n, hd = qkv.shape[0], heads * d
        assert qkv.shape[1] == (3 * hd if v_ext is None else 2 * hd)
        q, k, v, v_heads = _split(qkv, v_ext, heads, d)
        stats = K.attn_fwd_reduce(q, k, v, heads, v_heads, d)
        n_total = float(n) if n_override is None else float(n_override)
        if shard is not None:
            shard.all_reduce(stats)
            if n_override is None:
                n_total = float(shard.n_global)
        out, den, o_heads = K.attn_fwd_apply(q, v, stats, n_total, heads, v_heads, d)
        per_head = bool(per_head) and heads > 1
        ctx.save_for_backward(qkv, v_ext, out, den, o_heads, stats)
        ctx.meta = (heads, d, n_total, shard, per_head)
        return o_heads if per_head else out

    @staticmethod
    def backward(ctx, g):
        qkv, v_ext, out, den, o_heads, stats = ctx.saved_tensors
        heads, d, n_total, shard, per_head = ctx.meta
        g = _rows(g.contiguous())
        hd = heads * d
        q, k, v, v_heads = _split(qkv, v_ext, heads, d)
        o = out if heads == 1 else o_heads
        kw = {"per_head": True} if per_head else {}
        bstats = K.attn_bwd_reduce(q, g, o, den, heads, d, **kw)
        if shard is not None:
            shard.all_reduce(bstats)
        dqkv = torch.empty_like(qkv)
        dv_ext = torch.empty_like(v_ext) if v_ext is not None else None
        dv = dqkv[:, 2 * hd:] if v_ext is None else dv_ext
        K.attn_bwd_apply(q, k, v, g, o, den, stats, bstats, n_total, heads, v_heads, d,
                         dqkv[:, :hd], dqkv[:, hd:2 * hd], dv, **kw)
        return dqkv, dv_ext, None, None, None, None, None


def attention(qkv: torch.Tensor, v_ext: Optional[torch.Tensor], heads: int, d: int, shard=None,
              n_total=None, per_head: bool = False):
    """mean_h (qn S + N V)/(qn z + N) from fused projections; see _Attention.  `n_total` overrides
    the N of large/ours.py:133 (default: the number of rows, or the global count when sharded).
    per_head (H > 1): the per-head outputs [N, H*d] instead of their mean, differentiable."""
    return _Attention.apply(qkv, v_ext, heads, d, shard, n_total, per_head)


# ------------------------------------------------------------------------------------------------
# T3 + T4 fused: attention straight from the un-projected input (H = 1, query == source)
# ------------------------------------------------------------------------------------------------
class _MM(torch.autograd.Function):
    """a @ b for small fp32 matrices on sgf_gemm (csrc/gemm.hip), differentiable: the d x d algebra's products stay inside
    libsgf.so also where autograd differentiates it (DIFFormer's sum_v form, the test references)."""

    @staticmethod
    def forward(ctx, a, b):
        ctx.save_for_backward(a, b)
        return K.gemm(a, b)

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        ga = K.gemm(g, b.t()) if ctx.needs_input_grad[0] else None
        gb = K.gemm(a.t(), g) if ctx.needs_input_grad[1] else None
        return ga, gb


def mm(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """Matrix product of two 2-D tensors (views welcome: strides are passed through) on sgf_gemm."""
    return _MM.apply(a, b)


def _attn_h_small_packed(Gt, Wqk, Wv, n_total: float, sum_v: bool = False):
    """The d x d algebra of include/sgf.h (sgf_attn_h_*) on AUGMENTED operands, as differentiable torch ops whose products
    run on sgf_gemm (mm) — the form DIFFormer's sum_v numerator takes (no recipe's hot path); SGFormer's own numerator runs
    as sgf_attn_h_small_fwd / _bwd (csrc/attn_small.hip), whose formulation this is (tests/attn_algebra.py holds the
    term-by-term restatement both are checked against):
        Gt  = [[G, s], [s^T, n_rows]]   = ht^T ht for ht = [h | 1]          [(D + 1) x (D + 1)]
        Wqk = [[wq | bq], [wk | bk]]                                        [2 d x (D + 1)]
        Wv  = [wv | bv]                                                     [d x (D + 1)]
    so that  K^T V = (Wk~ Gt) Wv~^T  (all four bias terms included),  K^T 1 = (Wk~ Gt)[:, D],  ||Q||_F^2 = sum((Wq~ Gt) * Wq~)
    and one product  Wq~^T [K^T V | K^T 1]  carries  wq^T s0, bq s0, wq^T z0 and bq . z0  in its blocks.
    Returns (M [D, d], m [d], w [D], beta [1])."""
    d = Wv.shape[0]
    D = Gt.shape[0] - 1
    PG = mm(Wqk, Gt)                                       # [2 d, D + 1]
    ssq = (PG * Wqk).view(2, -1).sum(1)                  # ||Q||^2, ||K||^2
    c = torch.rsqrt(ssq[0] * ssq[1])                     # (not ssq.prod(): its backward looks for zeros on the HOST)
    PK = PG[d:]
    SZ = torch.cat([mm(PK, Wv.t()), PK[:, D:]], 1)          # [s0 | z0]   [d, d + 1]
    U = mm(Wqk[:d].t(), SZ) * c                          # rows :D = wq^T [s0 | z0], row D = bq [s0 | z0]
    if sum_v:
        Mm = torch.cat([U[:D, :d], (U[D, :d] + mm(Wv, Gt[:, D:]).view(-1))[None]], 0)
    else:
        Mm = torch.add(U[:, :d], Wv.t(), alpha=n_total)  # rows :D = M, row D = m
    wb = U[:, d].contiguous()
    return Mm[:D], Mm[D], wb[:D], wb[D:] + n_total


def _attn_h_pack(G, s, n_rows: float, wq, bq, wk, bk, wv, bv):
    """The augmented operands of _attn_h_small_packed (no autograd: the caller differentiates w.r.t. the packed leaves
    and slices their gradients)."""
    D = G.shape[0]
    Gt = torch.empty((D + 1, D + 1), dtype=_F32, device=G.device)
    Gt[:D, :D] = G
    Gt[:D, D] = s
    Gt[D, :D] = s
    Gt[D, D].fill_(n_rows)          # (NOT `= n_rows`: assigning a Python scalar is a host -> device copy that syncs the stream)
    Wqk = torch.cat([torch.cat([wq, wk], 0), torch.cat([bq, bk])[:, None]], 1)
    Wv = torch.cat([wv, bv[:, None]], 1)
    return Gt, Wqk, Wv


class _AttentionFromInput(torch.autograd.Function):
    """out = full_attention_conv(h Wq^T + bq, h Wk^T + bk, h Wv^T + bv) for ONE head, without ever
    materialising Q / K / V (include/sgf.h, "attention straight from the un-projected layer
    input").  wv / bv None = V is h itself (use_weight=False, large/ours.py:128)."""

    @staticmethod
    def forward(ctx, h, wq, bq, wk, bk, wv, bv, shard, n_override, sum_v=False, tap=None):
        K.check(h)
        h = _rows(h)
        n, d = h.shape
        ctx.tap = tap
        if tap is not None:
            tap["armed"] = tap["sibling"] = True   # a GradTap on the same input may hand its gradient to this node's backward
        f32 = [t.detach().float() for t in (wq, bq, wk, bk)]
        if wv is None:
            f32 += [torch.eye(d, dtype=_F32, device=h.device), torch.zeros(d, dtype=_F32, device=h.device)]
        else:
            f32 += [wv.detach().float(), bv.detach().float()]
        G, s = K.gram(h, h)                       # [d, d], [d]: the only global reduction of the forward
        n_rows = float(n)
        if shard is not None:
            gs = torch.cat([G.reshape(-1), s])
            shard.all_reduce(gs)
            G, s = gs[:d * d].reshape(d, d), gs[d * d:]
            n_rows = float(shard.n_global)
        n_total = n_rows if n_override is None else float(n_override)
        # the d x d algebra on packed operands, recorded ONCE: the backward differentiates this graph instead of re-running it
        di = f32[0].shape[1]
        if sum_v:           # DIFFormer's numerator: the autograd form (no recipe's hot path)
            with torch.enable_grad():
                leaves = [t.requires_grad_(True) for t in _attn_h_pack(G, s, n_rows, *f32)]
                small = _attn_h_small_packed(*leaves, n_total, sum_v=True)
            M, m, w, beta = (t.detach().contiguous() for t in small)
            ctx.small = (leaves, small, None)
        else:               # ONE library call each way (sgf_attn_h_small_fwd / _bwd: 6 + 9 launches, csrc/attn_small.hip)
            M, m, w, beta, saved = K.attn_h_small_fwd(G, s, n_rows, n_total, f32[0], f32[1], f32[2], f32[3],
                                                      None if wv is None else f32[4], None if wv is None else f32[5])
            ctx.small = (None, None, saved)
        out, den = K.attn_h_fwd(h, M, m, w, beta)
        ctx.save_for_backward(h, out, den, G, s, M, w, *f32)
        ctx.meta = (n_rows, n_total, shard, wv is None,
                    [None if t is None else t.dtype for t in (wq, bq, wk, bk, wv, bv)], bool(sum_v))
        return out

    @staticmethod
    def backward(ctx, g):
        h, out, den, G, s, M, w, *f32 = ctx.saved_tensors
        n_rows, n_total, shard, v_is_h, dtypes, sum_v = ctx.meta
        d = h.shape[1]
        g = _rows(g.contiguous())
        split = K.attn_h_bwd_split_supported(h, g, out)
        if split:   # the first apply pass needs only forward quantities and yields the row scalars the reduce wants
            rowscal = K.attn_h_bwd_pre(g, out, den, M, w)
            hstats = K.attn_h_bwd_reduce_scaled(h, g, rowscal)
        else:
            hstats = K.attn_h_bwd_reduce(h, g, out, den)      # [dM | dw | dm | dbeta]
        if shard is not None:
            shard.all_reduce(hstats)
        dM, dw_, dm = hstats[:d * d].reshape(d, d), hstats[d * d:d * d + d], hstats[d * d + d:d * d + 2 * d]
        dbeta = hstats[d * d + 2 * d:]
        # backward through the d x d algebra: the graph the forward recorded on the packed operands; the gradients of
        # G, s and the six parameters are blocks of the packed ones
        leaves, small, saved = ctx.small
        do, di = f32[0].shape
        if saved is None:       # DIFFormer's sum_v form: the graph the forward recorded (products on sgf_gemm through mm)
            gGt, gWqk, gWv = torch.autograd.grad(small, leaves, grad_outputs=(dM, dm, dw_, dbeta), retain_graph=True)
            Dm = gGt[:di, :di]
            D = (Dm + Dm.t()).contiguous()
            ds = gGt[:di, di] + gGt[di, :di]
            grads = (None, None, gWqk[:do, :di].contiguous(), gWqk[:do, di].contiguous(), gWqk[do:, :di].contiguous(),
                     gWqk[do:, di].contiguous(), gWv[:, :di].contiguous(), gWv[:, di].contiguous())
        else:
            D, ds, *pgr = K.attn_h_small_bwd(dM, dw_, dm, dbeta, n_total, saved, di, do, want_v=not v_is_h)
            grads = (None, None, *pgr)
        # the other gradient of h (the residual branch's, parked by a GradTap that ran before this node) is added in
        # the last pass instead of by autograd's separate three-tensor add
        extra = None
        if ctx.tap is not None:
            extra = ctx.tap.pop("grad", None)
            if extra is None:
                ctx.tap["armed"] = False     # the tap has not run yet: tell it to pass its gradient through
        if split:
            fold = extra is not None and extra.dtype == h.dtype and extra.shape == h.shape
            dh = K.attn_h_bwd_post(h, D, ds.contiguous(), _rows16(extra) if fold else None)
            if extra is not None and not fold:
                dh = dh + extra
        else:
            dh = K.attn_h_bwd_apply(h, g, out, den, M, w, D, ds.contiguous())
            if extra is not None:
                dh = dh + extra
        pg = list(grads[2:])
        if shard is not None:   # parameter grads are summed over ranks again by ShardContext.sync_grads
            pg = [None if t is None else shard.unsum(t) for t in pg]
        if v_is_h:
            pg[4] = pg[5] = None
        pg = [None if (t is None or dt is None) else t.to(dt) for t, dt in zip(pg, dtypes)]
        return (dh, *pg, None, None, None, None)


def attention_from_input(h, wq, bq, wk, bk, wv=None, bv=None, shard=None, n_total=None, sum_v=False, tap=None):
    """One-head linear attention from the un-projected input.  sum_v=True: DIFFormer's 'simple' kernel
    (numerator q S + sum_l V_l instead of q S + N V_n; medium/difformer.py:18-39).  tap: see grad_tap."""
    return _AttentionFromInput.apply(h, wq, bq, wk, bk, wv, bv, shard, n_total, sum_v, tap)


class _GradTap(torch.autograd.Function):
    """Identity.  Its backward PARKS the incoming gradient in `holder` when a sibling consumer of the same tensor
    (attention_from_input(..., tap=holder)) has armed it, and returns nothing itself: the sibling's last kernel adds
    the parked gradient to its own.  Works in either execution order of the two backward nodes (the engine runs the
    node created later first, so call grad_tap AFTER the sibling's forward): if the sibling already ran, the
    gradient passes through and autograd adds as usual."""

    @staticmethod
    def forward(ctx, x, holder):
        ctx.holder = holder
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        holder = ctx.holder
        if holder.get("armed"):
            holder["grad"] = g
            return None, None
        holder["armed"] = bool(holder.get("sibling"))    # the sibling ran first: pass through, re-arm for a next backward
        return g, None


def grad_tap(x, holder):
    """x, as the SECOND consumer of a tensor whose first consumer is attention_from_input(x, ..., tap=holder)
    (large/ours.py:206-208: the layer input feeds the attention and the residual)."""
    if not (torch.is_grad_enabled() and x.requires_grad):
        return x
    return _GradTap.apply(x, holder)


def attention_stats(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor):
    """Un-normalised partials [S0 | z0 | ssq_q | ssq_k] (test / inspection helper, no autograd)."""
    K.check(q, k, v)
    n, heads, d = q.shape
    q2, k2, v2 = (_rows(t.reshape(n, -1)) for t in (q, k, v))
    return K.attn_fwd_reduce(q2, k2, v2, heads, v.shape[1], d)


# ------------------------------------------------------------------------------------------------
# T5: y = [relu](LayerNorm(a*x + b*res))   (large/ours.py:198-202, 210-216)
# ------------------------------------------------------------------------------------------------
class _LNResAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, res, a: float, b: float, gamma, beta, relu: bool, eps: float):
        K.check(x, res, gamma)
        x, res = _rows(x), _rows(res)
        has_ln = gamma is not None
        g32 = gamma.detach().float().contiguous() if has_ln else None
        b32 = beta.detach().float().contiguous() if has_ln else None
        y, mean, rstd = K.ln_fwd(x, res, a, b, g32, b32, relu, eps)
        ctx.save_for_backward(x, res, y if relu else None, g32, mean, rstd)
        ctx.meta = (float(a), float(b), bool(relu), gamma.dtype if has_ln else None)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, res, y, g32, mean, rstd = ctx.saved_tensors
        a, b, relu, pdtype = ctx.meta
        gy = _rows(gy.contiguous())
        dx, dres, dgamma, dbeta = K.ln_bwd(gy, y, x, res, a, b, g32, relu, mean, rstd)
        if g32 is not None:
            dgamma, dbeta = dgamma.to(pdtype), dbeta.to(pdtype)
        return dx, dres, None, None, dgamma, dbeta, None, None


def ln_res_act(x, res, a, b, gamma, beta, relu, eps=1e-5):
    return _LNResAct.apply(x, res, a, b, gamma, beta, relu, eps)


# ------------------------------------------------------------------------------------------------
# T6: y = [relu](BatchNorm1d(x)) [+ res]   (large/ours.py:77-81, 87-93)
# ------------------------------------------------------------------------------------------------
class _BNActRes(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, res, gamma, beta, mean, rstd, relu: bool, training: bool, n_tot: float,
                shard):
        K.check(x, res)
        x, res = _rows(x), _rows(res)
        g32 = gamma.detach().float().contiguous() if gamma is not None else None
        b32 = beta.detach().float().contiguous() if beta is not None else None
        mean = mean.detach().float().contiguous()
        rstd = rstd.detach().float().contiguous()
        y = K.bn_apply(x, mean, rstd, g32, b32, res, relu)
        ctx.save_for_backward(x, g32, b32, mean, rstd)
        ctx.meta = (bool(relu), bool(training), float(n_tot), shard, res is not None,
                    gamma.dtype if gamma is not None else None)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, g32, b32, mean, rstd = ctx.saved_tensors
        relu, training, n_tot, shard, has_res, pdtype = ctx.meta
        gy = _rows(gy.contiguous())
        d = x.shape[1]
        stats = K.bn_bwd_stats(gy, x, mean, rstd, g32, b32, relu)
        if shard is not None:
            shard.all_reduce(stats)  # also makes dgamma / dbeta the global sums
        dx = K.bn_bwd_apply(gy, x, mean, rstd, g32, b32, relu, stats, 1.0 / max(n_tot, 1.0), training)
        dgamma = stats[d:].to(pdtype) if g32 is not None else None
        dbeta = stats[:d].to(pdtype) if b32 is not None else None
        if shard is not None and g32 is not None:
            # parameter grads are summed over ranks again by ShardContext.sync_grads: pre-divide
            dgamma, dbeta = shard.unsum(dgamma), shard.unsum(dbeta)
        return dx, (gy if has_res else None), dgamma, dbeta, None, None, None, None, None, None


def bn_act_res(x, res, gamma, beta, mean, rstd, relu, training, n_tot, shard=None):
    return _BNActRes.apply(x, res, gamma, beta, mean, rstd, relu, training, n_tot, shard)


# ------------------------------------------------------------------------------------------------
# dropout [+ residual]  (F.dropout at large/ours.py:81,92,202,216; the add of :93)
# ------------------------------------------------------------------------------------------------
class _DropoutRes(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, res, p: float):
        K.check(x, res)
        x, res = _rows(x), _rows(res)
        # a fresh 62-bit seed from torch's CPU generator: reproducible under torch.manual_seed, no sync
        seed = int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64))
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            # ranks seeded alike must not drop the same pattern in their row shards
            seed = (seed + (torch.distributed.get_rank() + 1) * 0x9E3779B97F4A7C15) % (2 ** 62)
        ctx.meta = (float(p), seed, res is not None)
        return K.dropout(x, res, p, seed)

    @staticmethod
    def backward(ctx, gy):
        p, seed, has_res = ctx.meta
        gy = _rows(gy.contiguous())
        return K.dropout(gy, None, p, seed), (gy if has_res else None), None


def dropout_res(x, res, p: float):
    """y = dropout(x, p) [+ res] in one pass, no stored mask (the backward recomputes it from the seed)."""
    return _DropoutRes.apply(x, res, p)


# ------------------------------------------------------------------------------------------------
# N4: the trainer's loss, log_softmax + NLLLoss on the training rows (large/main.py:139-141)
# ------------------------------------------------------------------------------------------------
class _NllRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels, idx, denom):
        K.check(logits, labels, idx)
        logits = logits if logits.stride(-1) == 1 else logits.contiguous()
        labels = labels.reshape(-1).contiguous()
        idx = idx.contiguous()
        if idx.dtype == torch.bool:
            idx = idx.nonzero().view(-1)
        m = idx.numel() if denom is None else denom
        ctx.save_for_backward(logits, labels, idx)
        ctx.inv = 1.0 / float(max(m, 1))
        return (K.nll_fwd(logits, labels, idx) * ctx.inv).reshape(())

    @staticmethod
    def backward(ctx, g):
        logits, labels, idx = ctx.saved_tensors
        return K.nll_bwd(logits, labels, idx, g.reshape(1).float().contiguous(), ctx.inv), None, None, None


def nll_loss_rows(logits, labels, idx, denom=None):
    """mean_j -log_softmax(logits[idx[j]])[labels[idx[j]]]  ==  the three lines large/main.py:139-141
    (`F.log_softmax` + `NLLLoss` on `out[train_idx]`) in one pass over the training rows; `denom`
    overrides the divisor (the GLOBAL training-row count of a node-sharded run).  fp32 math on fp32
    or bf16 logits; the gradient is written for all N rows (zeros off the training rows)."""
    return _NllRows.apply(logits, labels, idx, denom)


# ------------------------------------------------------------------------------------------------
# row permutation at the module boundary (re-ordered graphs) and mini-batch row gathers
# ------------------------------------------------------------------------------------------------
class _PermuteRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, idx, idx_inv, out_dtype):
        K.check(x, idx)
        ctx.save_for_backward(idx_inv)
        ctx.in_dtype = x.dtype
        return K.gather_rows(x, idx, out_dtype)

    @staticmethod
    def backward(ctx, g):
        (idx_inv,) = ctx.saved_tensors
        return K.gather_rows(g, idx_inv, ctx.in_dtype), None, None, None


def permute_rows(x: torch.Tensor, idx: torch.Tensor, idx_inv: torch.Tensor, out_dtype=None) -> torch.Tensor:
    """y[i] = x[idx[i]] for a PERMUTATION idx with inverse idx_inv (the gradient is then the gather by
    idx_inv — no scatter, no atomics); optional storage cast on the way (fp32 features -> bf16)."""
    return _PermuteRows.apply(x, idx, idx_inv, out_dtype)


def gather_rows(x: torch.Tensor, idx: torch.Tensor, out_dtype=None) -> torch.Tensor:
    """x[idx] on the GPU without autograd (mini-batch feature gather, large/main-batch.py:138)."""
    K.check(x, idx)
    return K.gather_rows(x, idx, out_dtype)


# ------------------------------------------------------------------------------------------------
# fan-out hub: one tensor, k consumers, ONE fused gradient sum (instead of k-1 pairwise ATen adds)
# ------------------------------------------------------------------------------------------------
class _FanOut(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, k: int):
        ctx.k = k
        return tuple(x.view_as(x) for _ in range(k))

    @staticmethod
    def backward(ctx, *grads):
        gs = [g for g in grads if g is not None]
        if not gs:
            return None, None
        if len(gs) == 1:
            return gs[0], None
        if gs[0].shape[1] % 4 != 0 or len(gs) > 8:
            out = gs[0]
            for g in gs[1:]:
                out = out + g
            return out, None
        return K.sum_n(gs), None


def fan_out(x: torch.Tensor, k: int):
    """k aliases of x whose gradients are summed in one pass (K.sum_n) in the backward."""
    if k <= 1 or not x.requires_grad:
        return tuple(x for _ in range(max(k, 1)))
    return _FanOut.apply(x, k)


# ------------------------------------------------------------------------------------------------
# T7: y = a*x1 + b*x2   (large/ours.py:269-270)
# ------------------------------------------------------------------------------------------------
class _Axpby(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x1, x2, a: float, b: float):
        K.check(x1, x2)
        ctx.ab = (float(a), float(b))
        return K.axpby(_rows(x1), a, _rows(x2), b)

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.ab
        ga = g * a
        return ga, (ga if a == b else g * b), None, None   # one pass when both weights are equal (gw = 0.5)


def axpby(x1, x2, a, b):
    return _Axpby.apply(x1, x2, a, b)


# ------------------------------------------------------------------------------------------------
# T7 fused: logits = (a x1 + b x2) W^T + bias   (large/ours.py:269-270 + :275 in one kernel)
# ------------------------------------------------------------------------------------------------
class _CombineFC(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x1, x2, w, bias, a: float, b: float, row_map=None):
        K.check(x1, x2)
        x1, x2 = _rows16(x1), _rows16(x2)        # sgf_combine_fc_* read 16-byte matrix-core fragments
        ctx.row_map = row_map
        w32 = w.detach().float().contiguous()
        b32 = bias.detach().float().contiguous()
        c = w32.shape[0]
        if (x1.dtype == _F32 or c > 64) and c % 4 != 0:   # the exact-fp32 kernel takes class counts % 4: zero rows, sliced off below
            w32 = torch.nn.functional.pad(w32, (0, 0, 0, 4 - c % 4))
            b32 = torch.nn.functional.pad(b32, (0, 4 - c % 4))
        ctx.save_for_backward(x1, x2, w32)
        ctx.meta = (float(a), float(b), w.dtype, bias.dtype, c)
        out = K.combine_fc_fwd(x1, a, x2, b, w32, b32) if row_map is None else K.combine_fc_fwd(x1, a, x2, b, w32, b32, row_map)
        return out if out.shape[1] == c else out[:, :c]

    @staticmethod
    def backward(ctx, g):
        x1, x2, w32 = ctx.saved_tensors
        a, b, wdtype, bdtype, c_true = ctx.meta
        g = g.float()
        if w32.shape[0] != g.shape[1]:           # fp32 storage: the padded class columns carry a zero gradient
            g = torch.nn.functional.pad(g, (0, w32.shape[0] - g.shape[1]))
        g = g.contiguous()
        row_map = ctx.row_map
        c = c_true
        if x1.dtype == _BF16 and hasattr(K, "combine_fc_bwd_g") and g.shape[1] <= 64:
            # the kernel holds the logits' gradient as bf16 matrix-core fragments anyway: it leaves them as the [N, 48]
            # operand of the weight gradient's node reductions (module row order), instead of a cast + a pad pass over g
            dx1, dx2, gp = K.combine_fc_bwd_g(g, w32, a, b, row_map)
            dw1, db = K.gram(gp, x1, want_colsum=True)
            dw2, _ = K.gram(gp, x2, want_colsum=False)
            dw = (a * dw1 + b * dw2)[:c]
            return dx1, dx2, dw.to(wdtype), db[:c].to(bdtype), None, None, None
        dx1, dx2 = K.combine_fc_bwd(g, w32, a, b, x1.dtype) if row_map is None else K.combine_fc_bwd(g, w32, a, b, x1.dtype,
                                                                                                       row_map)
        # dW = a g^T x1 + b g^T x2, db = colsum(g): node reductions on sgf_gram (g in the activation dtype, its
        # width padded to a multiple of 4 — the same rounding the unfused path applies to the logits gradient)
        if row_map is not None:
            # g is in the CALLER's row order, x1 / x2 in the module's: one gather pass brings g over, cast included
            gp = K.gather_rows(g, row_map, x1.dtype)
        else:
            gp = g.to(x1.dtype)
        if gp.shape[1] % 4 != 0:
            gp = torch.nn.functional.pad(gp, (0, 4 - gp.shape[1] % 4))
        gp = _rows(gp)
        dw1, db = K.gram(gp, x1, want_colsum=True)
        dw2, _ = K.gram(gp, x2, want_colsum=False)
        dw = (a * dw1 + b * dw2)[:c]
        return dx1, dx2, dw.to(wdtype), db[:c].to(bdtype), None, None, None


def combine_fc(x1, x2, w, bias, a: float, b: float, row_map=None) -> torch.Tensor:
    """fp32 logits = (a x1 + b x2) W^T + bias without materialising the combination (sgf_combine_fc_*).  row_map (int32
    permutation, bf16 storage with at most 64 classes: combine_fc_mapped_supported): the logits leave in the caller's row
    order — row j of the product is row row_map[j] of the result — and the backward reads the incoming gradient through the
    same map, instead of two [N, C] gather passes around the head."""
    return _CombineFC.apply(x1, x2, w, bias, a, b, row_map)


def combine_fc_mapped_supported(x: torch.Tensor, classes: int) -> bool:
    import os
    return (x.dim() == 2 and hasattr(K, "combine_fc_mapped_supported") and os.environ.get("SGF_HEAD_MAPPED", "1") != "0"
            and K.combine_fc_mapped_supported(x.shape[1], classes, x.dtype))


def combine_fc_supported(x: torch.Tensor, classes: int) -> bool:
    import os
    if x.dim() == 2 and x.dtype == _BF16 and classes > 64 and os.environ.get("SGF_HEAD_WIDE", "0") != "1":
        # bf16 storage with more than 64 classes: sgf_combine_fc_* would run on the exact-fp32 matrix cores (157 TF, padded
        # to 256 x 256) and is compute-bound there — measured on the 100M recipe (d = 128, C = 172, N = 13.9 M): 18.5 + 16.3 ms
        # per step against ~12 ms for sgf_axpby + the bf16 streaming Linear, 251.9 vs 234.6 ms per step.  The logits
        # (N x C fp32) dominate that head's traffic either way; the one-kernel form stays available (SGF_HEAD_WIDE=1).
        return False
    return x.dim() == 2 and K.combine_fc_supported(x.shape[1], classes, x.dtype)



# ------------------------------------------------------------------------------------------------
# T4 / T6: the Linear layers, the stems and the GraphConv layer live in linear.py; every name is available here
# ------------------------------------------------------------------------------------------------
from . import linear as _linear  # noqa: E402
from .linear import *  # noqa: F401,F403,E402
from .linear import (_BN_SAMPLE_ROWS, _Linear, _LinearBNActRes, _StemPair, _StemPairBN, _acc_in_place, _fused_bwd, _linear_param_grads,  # noqa: F401,E402
                     _linear_with_stats, _streaming_linear, _streaming_linear_ok)
