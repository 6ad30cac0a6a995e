"""Autograd-level operators over the libsgf C ABI (include/sgf.h).

Each operator is the reference arithmetic of one row of SURVEY.md §8a, forward and backward, as a
`torch.autograd.Function` whose forward/backward call the HIP kernels on the current stream.
PyTorch is used for memory, streams and autograd bookkeeping only; there is no eager fallback — a
CPU tensor or a missing library raises.

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

_F32 = torch.float32
_BF16 = torch.bfloat16


# ------------------------------------------------------------------------------------------------
# plumbing
# ------------------------------------------------------------------------------------------------
def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream(device) -> ctypes.c_void_p:
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _code(t: torch.Tensor) -> int:
    if t.dtype == _F32:
        return _lib.SGF_F32
    if t.dtype == _BF16:
        return _lib.SGF_BF16
    raise TypeError(f"sgformer_amd kernels take float32 or bfloat16 storage, got {t.dtype}")


def _require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError(
                "sgformer_amd runs on MI355X only: got a CPU tensor.  There is no CPU fallback; "
                "move the model and its inputs to the GPU (the reference's CPU evaluation path, "
                "large/eval.py:36-65, is outside this library's scope).")


def _rows(t: torch.Tensor) -> torch.Tensor:
    """Return a view/copy of a 2-D tensor whose rows are contiguous and 4-element aligned."""
    if t.stride(-1) != 1 or t.stride(0) % 4 != 0 or t.data_ptr() % (4 * t.element_size()) != 0:
        t = t.contiguous()
        if t.stride(0) % 4 != 0:
            raise ValueError(f"feature dimension {t.shape[-1]} must be a multiple of 4")
    return t


_workspaces: "dict[tuple, torch.Tensor]" = {}


def _workspace(device, name: str, nbytes: int) -> torch.Tensor:
    """Per-device scratch reused across calls (all users run on the current stream, in order)."""
    key = (device.index, name, torch.cuda.current_stream(device).cuda_stream)
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
        _workspaces[key] = ws
    return ws


# ------------------------------------------------------------------------------------------------
# T1: cached CSR of the normalised adjacency (large/ours.py:26-33)
# ------------------------------------------------------------------------------------------------
class CSRGraph:
    """rowptr/colind/val of A = D^-1/2 (edge_index^T) D^-1/2 on the GPU, built by sgf_csr_build.

    The reference rebuilds this (degree + argsort over nnz) in every layer of every forward;
    here it is built once per `edge_index` and reused by all layers and by the backward.
    """

    def __init__(self, edge_index: torch.Tensor, num_nodes: int, validate: bool = True):
        _require_cuda(edge_index)
        if edge_index.dtype != torch.int64 or edge_index.dim() != 2 or edge_index.shape[0] != 2:
            raise ValueError("edge_index must be an int64 tensor of shape [2, nnz]")
        if num_nodes >= 2 ** 31:
            raise ValueError("num_nodes must be < 2^31 (int32 column indices)")
        ei = edge_index.contiguous()
        dev = ei.device
        nnz = int(ei.shape[1])
        n = int(num_nodes)
        if validate and nnz > 0:
            lo, hi = torch.aminmax(ei)
            if int(lo) < 0 or int(hi) >= n:
                raise IndexError(f"edge_index has node ids outside [0, {n})")
        self.n, self.nnz, self.device = n, nnz, dev
        self.edge_index = ei
        self.rowptr = torch.empty(n + 1, dtype=torch.int64, device=dev)
        self.colind = torch.empty(nnz, dtype=torch.int32, device=dev)
        self.val = torch.empty(nnz, dtype=_F32, device=dev)
        self.deg = torch.empty(n, dtype=torch.int32, device=dev)
        lib = _lib.load()
        nbytes = lib.sgf_csr_workspace_bytes(nnz, n)
        ws = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            _lib.call("sgf_csr_build", _ptr(ei), nnz, n, _ptr(self.rowptr), _ptr(self.colind),
                      _ptr(self.val), _ptr(self.deg), _ptr(ws), ws.numel(), _stream(dev))
        self._t = None  # (rowptr, colind, val) of A^T, built on first backward
        self.symmetric: Optional[bool] = None

    def transposed(self):
        """CSR of A^T for dX = A^T dY; the same arrays when A is symmetric (one host sync, once)."""
        if self._t is None:
            dev, n, nnz = self.device, self.n, self.nnz
            t_rowptr = torch.empty(n + 1, dtype=torch.int64, device=dev)
            t_colind = torch.empty(nnz, dtype=torch.int32, device=dev)
            t_val = torch.empty(nnz, dtype=_F32, device=dev)
            flag = torch.zeros(1, dtype=torch.int32, device=dev)
            lib = _lib.load()
            nbytes = lib.sgf_csr_workspace_bytes(nnz, n)
            ws = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=dev)
            with torch.cuda.device(dev):
                _lib.call("sgf_csr_transpose", _ptr(self.edge_index), nnz, n, _ptr(self.deg),
                          _ptr(self.rowptr), _ptr(self.colind), _ptr(t_rowptr), _ptr(t_colind),
                          _ptr(t_val), _ptr(flag), _ptr(ws), ws.numel(), _stream(dev))
            self.symmetric = bool(int(flag.item()))
            if self.symmetric:
                self._t = (self.rowptr, self.colind, self.val)
            else:
                self._t = (t_rowptr, t_colind, t_val)
        return self._t


class _GraphCache:
    """edge_index -> CSRGraph, keyed on tensor identity AND version: the
    mini-batch trainers hand in a fresh edge_index every step, so entries are bounded (LRU) and each
    entry pins its key tensor so a recycled data_ptr can never alias a stale graph."""

    def __init__(self, capacity: int = 4):
        self.capacity = capacity
        self._d: "OrderedDict[tuple, CSRGraph]" = OrderedDict()

    def get(self, edge_index: torch.Tensor, num_nodes: int) -> CSRGraph:
        key = (edge_index.data_ptr(), edge_index._version, tuple(edge_index.shape),
               tuple(edge_index.stride()), str(edge_index.device), int(num_nodes))
        g = self._d.get(key)
        if g is not None:  # the entry pins its tensor, so an equal key means the same live memory
            self._d.move_to_end(key)
            return g
        g = CSRGraph(edge_index, num_nodes)
        g.edge_index_ref = edge_index
        self._d[key] = g
        while len(self._d) > self.capacity:
            self._d.popitem(last=False)
        return g

    def clear(self):
        self._d.clear()


graph_cache = _GraphCache()


# ------------------------------------------------------------------------------------------------
# T2: SpMM (large/ours.py:34)
# ------------------------------------------------------------------------------------------------
def _spmm_raw(rowptr, colind, val, x: torch.Tensor, n_rows: int) -> torch.Tensor:
    x = _rows(x)
    d = x.shape[1]
    y = torch.empty((n_rows, d), dtype=x.dtype, device=x.device)
    if n_rows == 0 or d == 0:
        return y
    with torch.cuda.device(x.device):
        _lib.call("sgf_spmm", _ptr(rowptr), _ptr(colind), _ptr(val), _ptr(x), x.stride(0), _ptr(y),
                  y.stride(0), n_rows, d, _code(x), _stream(x.device))
    return y


class _SpMM(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, graph: CSRGraph, shard):
        _require_cuda(x)
        ctx.graph, ctx.shard = graph, shard
        if shard is not None:
            # node-sharded: rows of A local, X rows gathered from all ranks (halo all-gather)
            xg = shard.all_gather_rows(x)
            return _spmm_raw(graph.rowptr, graph.colind, graph.val, xg, graph.n_local)
        return _spmm_raw(graph.rowptr, graph.colind, graph.val, x, graph.n)

    @staticmethod
    def backward(ctx, gy):
        graph, shard = ctx.graph, ctx.shard
        if shard is not None:
            # dX_local = (A^T dY)[local rows]; A symmetric => A^T's local rows = local rows of A
            gyg = shard.all_gather_rows(gy.contiguous())
            rp, ci, va = graph.transposed_local()
            return _spmm_raw(rp, ci, va, gyg, graph.n_local), None, None
        rp, ci, va = graph.transposed()
        return _spmm_raw(rp, ci, va, gy.contiguous(), graph.n), None, None


def spmm(graph: CSRGraph, x: torch.Tensor, shard=None) -> torch.Tensor:
    """Y = A X with A the cached normalised adjacency (torch_sparse.matmul(adj, x) in the reference)."""
    return _SpMM.apply(x, graph, shard)


# ------------------------------------------------------------------------------------------------
# T3: linear global attention (large/ours.py:130-157)
# ------------------------------------------------------------------------------------------------
class _Attention(torch.autograd.Function):
    """qk: [N, 2*H*d] = [Q | K] (or qkv: [N, 3*H*d] = [Q | K | V]); v: [N, d] when V is not projected."""

    @staticmethod
    def forward(ctx, qkv, v_ext, heads: int, d: int, shard, n_override=None):
        _require_cuda(qkv, v_ext)
        qkv = _rows(qkv)
        n = qkv.shape[0]
        hd = heads * d
        dev = qkv.device
        if v_ext is None:
            assert qkv.shape[1] == 3 * hd
            v, ldv, v_heads = qkv[:, 2 * hd:], qkv.stride(0), heads
        else:
            assert qkv.shape[1] == 2 * hd
            v_ext = _rows(v_ext)
            v, ldv, v_heads = v_ext, v_ext.stride(0), 1
        q, k = qkv[:, :hd], qkv[:, hd:2 * hd]
        ld = qkv.stride(0)
        lib = _lib.load()
        code = _code(qkv)
        slen = lib.sgf_attn_stats_len(heads, d)
        stats = torch.empty(slen, dtype=_F32, device=dev)
        ws = _workspace(dev, "attn", lib.sgf_attn_workspace_bytes(n, heads, d))
        st = _stream(dev)
        n_total = float(n) if n_override is None else float(n_override)
        with torch.cuda.device(dev):
            _lib.call("sgf_attn_fwd_reduce", _ptr(q), ld, _ptr(k), ld, _ptr(v), ldv, n, heads,
                      v_heads, d, code, _ptr(stats), _ptr(ws), ws.numel(), st)
            if shard is not None:
                shard.all_reduce(stats)
                if n_override is None:
                    n_total = float(shard.n_global)
            out = torch.empty((n, d), dtype=qkv.dtype, device=dev)
            den = torch.empty((n, heads), dtype=_F32, device=dev)
            o_heads = torch.empty((n, heads, d), dtype=qkv.dtype, device=dev) if heads > 1 else None
            _lib.call("sgf_attn_fwd_apply", _ptr(q), ld, _ptr(v), ldv, n, n_total, heads, v_heads, d,
                      code, _ptr(stats), _ptr(out), out.stride(0), _ptr(den), _ptr(o_heads), st)
        ctx.save_for_backward(qkv, v_ext, out, den, o_heads, stats)
        ctx.meta = (heads, d, n_total, shard)
        return out

    @staticmethod
    def backward(ctx, g):
        qkv, v_ext, out, den, o_heads, stats = ctx.saved_tensors
        heads, d, n_total, shard = ctx.meta
        g = _rows(g.contiguous())
        n = qkv.shape[0]
        hd = heads * d
        dev = qkv.device
        ld = qkv.stride(0)
        q, k = qkv[:, :hd], qkv[:, hd:2 * hd]
        if v_ext is None:
            v, ldv, v_heads = qkv[:, 2 * hd:], ld, heads
        else:
            v, ldv, v_heads = v_ext, v_ext.stride(0), 1
        o, ldo = (out, out.stride(0)) if heads == 1 else (o_heads, hd)
        lib = _lib.load()
        code = _code(qkv)
        blen = lib.sgf_attn_bstats_len(heads, d)
        bstats = torch.empty(blen, dtype=_F32, device=dev)
        ws = _workspace(dev, "attn", lib.sgf_attn_workspace_bytes(n, heads, d))
        st = _stream(dev)
        dqkv = torch.empty_like(qkv)
        dq, dk = dqkv[:, :hd], dqkv[:, hd:2 * hd]
        if v_ext is None:
            dv, lddv, dv_ext = dqkv[:, 2 * hd:], dqkv.stride(0), None
        else:
            dv_ext = torch.empty_like(v_ext)
            dv, lddv = dv_ext, dv_ext.stride(0)
        with torch.cuda.device(dev):
            _lib.call("sgf_attn_bwd_reduce", _ptr(q), ld, _ptr(g), g.stride(0), _ptr(o), ldo,
                      _ptr(den), n, heads, d, code, _ptr(bstats), _ptr(ws), ws.numel(), st)
            if shard is not None:
                shard.all_reduce(bstats)
            _lib.call("sgf_attn_bwd_apply", _ptr(q), ld, _ptr(k), ld, _ptr(v), ldv, _ptr(g),
                      g.stride(0), _ptr(o), ldo, _ptr(den), n, n_total, heads, v_heads, d, code,
                      _ptr(stats), _ptr(bstats), _ptr(dq), dqkv.stride(0), _ptr(dk), dqkv.stride(0),
                      _ptr(dv), lddv, st)
        return dqkv, dv_ext, None, None, None, None


def attention(qkv: torch.Tensor, v_ext: Optional[torch.Tensor], heads: int, d: int, shard=None,
              n_total=None):
    """mean_h (qn S + N V)/(qn z + N) from fused projections; see _Attention.  `n_total` overrides
    the N of large/ours.py:133 (default: the number of rows, or the global count when sharded)."""
    return _Attention.apply(qkv, v_ext, heads, d, shard, n_total)


def attention_stats(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor):
    """Un-normalised partials [S0 | z0 | ssq_q | ssq_k] (test / inspection helper, no autograd)."""
    _require_cuda(q, k, v)
    n, heads, d = q.shape
    v_heads = v.shape[1]
    q2, k2, v2 = (_rows(t.reshape(n, -1)) for t in (q, k, v))
    lib = _lib.load()
    stats = torch.empty(lib.sgf_attn_stats_len(heads, d), dtype=_F32, device=q.device)
    ws = _workspace(q.device, "attn", lib.sgf_attn_workspace_bytes(n, heads, d))
    with torch.cuda.device(q.device):
        _lib.call("sgf_attn_fwd_reduce", _ptr(q2), q2.stride(0), _ptr(k2), k2.stride(0), _ptr(v2),
                  v2.stride(0), n, heads, v_heads, d, _code(q2), _ptr(stats), _ptr(ws), ws.numel(),
                  _stream(q.device))
    return stats


# ------------------------------------------------------------------------------------------------
# T5: y = [relu](LayerNorm(a*x + b*res))   (large/ours.py:198-202, 210-216)
# ------------------------------------------------------------------------------------------------
class _LNResAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, res, a: float, b: float, gamma, beta, relu: bool, eps: float):
        _require_cuda(x, res, gamma)
        x = _rows(x)
        res = None if res is None else _rows(res)
        n, d = x.shape
        dev = x.device
        y = torch.empty((n, d), dtype=x.dtype, device=dev)
        has_ln = gamma is not None
        g32 = gamma.float().contiguous() if has_ln else None
        b32 = beta.float().contiguous() if has_ln else None
        mean = torch.empty(n, dtype=_F32, device=dev) if has_ln else None
        rstd = torch.empty(n, dtype=_F32, device=dev) if has_ln else None
        with torch.cuda.device(dev):
            _lib.call("sgf_ln_fwd", _ptr(x), x.stride(0), _ptr(res), 0 if res is None else res.stride(0),
                      float(a), float(b), _ptr(g32), _ptr(b32), int(relu), float(eps), n, d, _code(x),
                      _ptr(y), y.stride(0), _ptr(mean), _ptr(rstd), _stream(dev))
        ctx.save_for_backward(x, res, y if relu else None, g32, mean, rstd)
        ctx.meta = (float(a), float(b), bool(relu), gamma.dtype if has_ln else None)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, res, y, g32, mean, rstd = ctx.saved_tensors
        a, b, relu, pdtype = ctx.meta
        gy = _rows(gy.contiguous())
        n, d = x.shape
        dev = x.device
        has_ln = g32 is not None
        dx = torch.empty_like(x)
        dres = torch.empty_like(res) if res is not None else None
        dgamma = torch.empty(d, dtype=_F32, device=dev) if has_ln else None
        dbeta = torch.empty(d, dtype=_F32, device=dev) if has_ln else None
        lib = _lib.load()
        ws = _workspace(dev, "ln", lib.sgf_ln_bwd_workspace_bytes(n, d))
        with torch.cuda.device(dev):
            _lib.call("sgf_ln_bwd", _ptr(gy), gy.stride(0), _ptr(y), 0 if y is None else y.stride(0),
                      _ptr(x), x.stride(0), _ptr(res), 0 if res is None else res.stride(0), a, b,
                      _ptr(g32), int(relu), _ptr(mean), _ptr(rstd), n, d, _code(x), _ptr(dx),
                      dx.stride(0), _ptr(dres), 0 if dres is None else dres.stride(0), _ptr(dgamma),
                      _ptr(dbeta), _ptr(ws), ws.numel(), _stream(dev))
        if has_ln:
            dgamma, dbeta = dgamma.to(pdtype), dbeta.to(pdtype)
        return dx, dres, None, None, dgamma, dbeta, None, None


def ln_res_act(x, res, a, b, gamma, beta, relu, eps=1e-5):
    return _LNResAct.apply(x, res, a, b, gamma, beta, relu, eps)


# ------------------------------------------------------------------------------------------------
# T6: y = [relu](BatchNorm1d(x)) [+ res]   (large/ours.py:77-81, 87-93)
# ------------------------------------------------------------------------------------------------
def _colstats(x: torch.Tensor, shift: Optional[torch.Tensor]) -> torch.Tensor:
    n, d = x.shape
    dev = x.device
    lib = _lib.load()
    stats = torch.empty(2 * d, dtype=_F32, device=dev)
    ws = _workspace(dev, "col", lib.sgf_colstats_workspace_bytes(n, d))
    with torch.cuda.device(dev):
        _lib.call("sgf_colstats", _ptr(x), x.stride(0), _ptr(shift), n, d, _code(x), _ptr(stats),
                  _ptr(ws), ws.numel(), _stream(dev))
    return stats


def batch_stats(x: torch.Tensor, shard=None):
    """Two-pass column mean / biased variance over all rows (all ranks when sharded)."""
    x = _rows(x)
    n, d = x.shape
    s1 = _colstats(x, None)[:d]
    n_tot = float(n)
    if shard is not None:
        s1 = s1.contiguous()
        shard.all_reduce(s1)
        n_tot = float(shard.n_global)
    mean = s1 / n_tot
    s2 = _colstats(x, mean.contiguous())[d:]
    if shard is not None:
        s2 = s2.contiguous()
        shard.all_reduce(s2)
    var = s2 / n_tot
    return mean, var, n_tot


class _BNActRes(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, res, gamma, beta, mean, rstd, relu: bool, training: bool, n_tot: float,
                shard):
        _require_cuda(x, res)
        x = _rows(x)
        res = None if res is None else _rows(res)
        n, d = x.shape
        dev = x.device
        g32 = gamma.float().contiguous() if gamma is not None else None
        b32 = beta.float().contiguous() if beta is not None else None
        mean = mean.float().contiguous()
        rstd = rstd.float().contiguous()
        y = torch.empty((n, d), dtype=x.dtype, device=dev)
        with torch.cuda.device(dev):
            _lib.call("sgf_bn_apply", _ptr(x), x.stride(0), _ptr(mean), _ptr(rstd), _ptr(g32), _ptr(b32),
                      _ptr(res), 0 if res is None else res.stride(0), int(relu), n, d, _code(x),
                      _ptr(y), y.stride(0), _stream(dev))
        ctx.save_for_backward(x, g32, b32, mean, rstd)
        ctx.meta = (bool(relu), bool(training), float(n_tot), shard, res is not None,
                    gamma.dtype if gamma is not None else None)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, g32, b32, mean, rstd = ctx.saved_tensors
        relu, training, n_tot, shard, has_res, pdtype = ctx.meta
        gy = _rows(gy.contiguous())
        n, d = x.shape
        dev = x.device
        lib = _lib.load()
        stats = torch.empty(2 * d, dtype=_F32, device=dev)
        ws = _workspace(dev, "col", lib.sgf_colstats_workspace_bytes(n, d))
        dx = torch.empty_like(x)
        with torch.cuda.device(dev):
            _lib.call("sgf_bn_bwd_stats", _ptr(gy), gy.stride(0), _ptr(x), x.stride(0), _ptr(mean),
                      _ptr(rstd), _ptr(g32), _ptr(b32), int(relu), n, d, _code(x), _ptr(stats), _ptr(ws),
                      ws.numel(), _stream(dev))
            if shard is not None:
                shard.all_reduce(stats)  # also makes dgamma / dbeta the global sums
            _lib.call("sgf_bn_bwd_apply", _ptr(gy), gy.stride(0), _ptr(x), x.stride(0), _ptr(mean),
                      _ptr(rstd), _ptr(g32), _ptr(b32), int(relu), _ptr(stats), 1.0 / max(n_tot, 1.0),
                      int(training), n, d, _code(x), _ptr(dx), dx.stride(0), _stream(dev))
        dgamma = stats[d:].to(pdtype) if g32 is not None else None
        dbeta = stats[:d].to(pdtype) if b32 is not None else None
        if shard is not None and g32 is not None:
            # parameter grads are all-reduced (averaged) again by the shard's grad sync: pre-divide
            dgamma, dbeta = shard.unsum(dgamma), shard.unsum(dbeta)
        return dx, (gy if has_res else None), dgamma, dbeta, None, None, None, None, None, None


def bn_act_res(x, res, gamma, beta, mean, rstd, relu, training, n_tot, shard=None):
    return _BNActRes.apply(x, res, gamma, beta, mean, rstd, relu, training, n_tot, shard)


# ------------------------------------------------------------------------------------------------
# T7: y = a*x1 + b*x2   (large/ours.py:269-270)
# ------------------------------------------------------------------------------------------------
class _Axpby(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x1, x2, a: float, b: float):
        _require_cuda(x1, x2)
        x1, x2 = _rows(x1), _rows(x2)
        n, d = x1.shape
        y = torch.empty((n, d), dtype=x1.dtype, device=x1.device)
        with torch.cuda.device(x1.device):
            _lib.call("sgf_axpby", _ptr(x1), x1.stride(0), float(a), _ptr(x2), x2.stride(0), float(b),
                      n, d, _code(x1), _ptr(y), y.stride(0), _stream(x1.device))
        ctx.ab = (float(a), float(b))
        return y

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.ab
        return g * a, g * b, None, None


def axpby(x1, x2, a, b):
    return _Axpby.apply(x1, x2, a, b)
