"""The graph side of the autograd-level operators (split out of ops.py in r06): the cached CSR of an edge list (T1,
large/ours.py:26-33), its locality-restoring node order and tile / row-block plans, and T2 — the sum-reduce SpMM with its
A^T backward (large/ours.py:34) on whichever kernel family the graph's structure selects.  `ops` re-exports every name."""
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


K = HipKernels()

# Rows with more stored entries than this are reduced by whole workgroups, segment by segment
# (sgf_spmm_split): one wave walking a 17 k-entry hub row of a power-law graph is a latency-bound tail.
# LONG_ROW / _SEGMENT: the long-row threshold and segment length of the SpMM kernels (sgformer_amd/kernels.py)


# Graphs with fewer stored entries than this are per-batch graphs (mini-batch trainers build one per step): their
# long-row segment count is not read back from the device (a host sync per batch) but bounded from nnz alone.
SMALL_GRAPH_NNZ = 1 << 23


def long_row_segments(rowptr: torch.Tensor, nnz: Optional[int] = None, max_row_len: Optional[int] = None) -> int:
    """sum over rows longer than LONG_ROW of ceil(len / segment): the `long_segments` argument of
    sgf_spmm_split.  One tiny device reduction + host read per CSR (done once, when it is built) — or, for
    small graphs whose nnz is known on the host, the bound  sum ceil(len/seg) <= nnz/seg + nnz/(LONG_ROW+1)
    without any device read (0 when no row can be long at all).  `max_row_len`: a bound on the longest row the caller
    GUARANTEES without looking (a sampled batch: its largest fan-out; NOT the node count — duplicate edges are kept) — at most LONG_ROW means no
    split path at all (no memset, no extra launches) for that CSR."""
    if rowptr.numel() <= 1:
        return 0
    if max_row_len is not None and max_row_len <= LONG_ROW:
        return 0
    if nnz is not None and nnz < SMALL_GRAPH_NNZ:
        return 0 if nnz <= LONG_ROW else nnz // _SEGMENT + nnz // (LONG_ROW + 1) + 1
    lens = rowptr[1:] - rowptr[:-1]
    segs = torch.where(lens > LONG_ROW, (lens + (_SEGMENT - 1)) // _SEGMENT, torch.zeros_like(lens))
    return int(segs.sum())


# ------------------------------------------------------------------------------------------------
# T1: cached CSR of the normalised adjacency (large/ours.py:26-33)
# ------------------------------------------------------------------------------------------------
class CSRGraph:
    """rowptr/colind/val of A = D^-1/2 (edge_index^T) D^-1/2 on the GPU, built by sgf_csr_build.

    The reference rebuilds this (degree + argsort over nnz) in every layer of every forward;
    here it is built once per `edge_index` and reused by all layers and by the backward.
    """

    def __init__(self, edge_index: torch.Tensor, num_nodes: int, validate: bool = True):
        K.check(edge_index)
        if edge_index.dtype != torch.int64 or edge_index.dim() != 2 or edge_index.shape[0] != 2:
            raise ValueError("edge_index must be an int64 tensor of shape [2, nnz]")
        if num_nodes >= 2 ** 31:
            raise ValueError("num_nodes must be < 2^31 (int32 column indices)")
        ei = edge_index.contiguous()
        n = int(num_nodes)
        if getattr(edge_index, "_sgf_trusted", False):
            validate = False      # produced by batching.subgraph / graph_prologue: ids are in range by construction
        if validate and ei.shape[1] > 0:
            lo, hi = torch.aminmax(ei)
            if int(lo) < 0 or int(hi) >= n:
                raise IndexError(f"edge_index has node ids outside [0, {n})")
        self.n, self.nnz, self.device = n, int(ei.shape[1]), ei.device
        self.edge_index = ei
        pre = getattr(edge_index, "_sgf_csr", None)
        if pre is not None and pre[0].numel() == n + 1 and pre[1].numel() == self.nnz:
            # the producer of this edge list (batching.subgraph on the parent's CSR: sgf_subgraph_csr_*) built the normalised
            # CSR in the same pass — bit for bit what sgf_csr_build would return for it (tests/test_gpu_r05.py)
            self.rowptr, self.colind, self.val, self.deg = pre
        else:
            self.rowptr, self.colind, self.val, self.deg = K.csr_build(ei, n)
        # longest possible row: only a bound the caller GUARANTEES (sampling.NeighborSampler marks its batches with their
        # largest fan-out).  The node count is no bound: sgf_csr_build keeps duplicate edges (large/ours.py:33 does not
        # coalesce), so a row of a small multigraph can exceed LONG_ROW entries — without a hint the nnz-based
        # bound (small graphs) or the exact count from rowptr applies.
        hint = getattr(edge_index, "_sgf_max_in_degree", None)
        self.long_segments = long_row_segments(self.rowptr, self.nnz, None if hint is None else int(hint))
        self.t_long_segments = 0
        self._t = None  # (rowptr, colind, val) of A^T, built on first backward
        self.symmetric: Optional[bool] = None
        if getattr(edge_index, "_sgf_symmetric", False):
            # an induced subgraph of a graph whose A^T == A was verified once (batching.subgraph): symmetric by construction —
            # no second sort, no comparison pass and no host read per batch in the first backward
            self.symmetric, self._t, self.t_long_segments = True, (self.rowptr, self.colind, self.val), self.long_segments

    def transposed(self):
        """CSR of A^T for dX = A^T dY; the same arrays when A is symmetric."""
        if self._t is None:
            t_rowptr, t_colind, t_val, sym = K.csr_transpose(self.edge_index, self.n, self.deg,
                                                             self.rowptr, self.colind)
            self.symmetric = sym
            self._t = (self.rowptr, self.colind, self.val) if sym else (t_rowptr, t_colind, t_val)
            self.t_long_segments = self.long_segments if sym else long_row_segments(t_rowptr, self.nnz)
        return self._t

    # ---- LDS-staged row-block SpMM: one plan per (orientation, storage dtype) ----
    blocked = False          # set by GraphView when the plan serves enough entries from LDS

    def plan(self, dtype, transposed: bool = False):
        """BlockedPlan for SpMMs with `dtype` storage on this CSR (or its transpose), or None."""
        if not self.blocked:
            return None
        if not hasattr(self, "_plans"):
            self._plans = {}
        if transposed:
            self.transposed()
            if self.symmetric:
                transposed = False
        key = (bool(transposed), dtype)
        if key not in self._plans:
            rp, ci, va = self.transposed() if transposed else (self.rowptr, self.colind, self.val)
            self._plans[key] = BlockedPlan(rp, ci, va, self.n, dtype)
        return self._plans[key]

    # ---- dense matrix-core tiles + gather remainder (sgf_spmm_tile): one plan per orientation ----
    tiled = False            # set by GraphView when the tile plan covers enough of the stored entries
    blk_row = None           # int32 [nb + 1] row blocks that follow the communities (GraphView)

    def tile_plan(self, transposed: bool = False):
        if not hasattr(self, "_tile_plans"):
            self._tile_plans = {}
        if transposed:
            self.transposed()
            if self.symmetric:
                transposed = False
        if transposed not in self._tile_plans:
            rp, ci, va = self.transposed() if transposed else (self.rowptr, self.colind, self.val)
            self._tile_plans[transposed] = TilePlan(rp, ci, va, self.n, self.blk_row,
                                                   min_count=getattr(self, "tile_min_count", None))
        return self._tile_plans[transposed]

    def view(self, now: bool = False) -> "GraphView":
        """How the model should run on this graph: the graph itself, or a re-ordered copy + the row
        permutation to apply at the module boundary (decided once; see GraphView).  now=True: decide
        at this call instead of waiting for the second forward."""
        self.forward_calls = getattr(self, "forward_calls", 0) + 1
        v = getattr(self, "_view", None)
        if v is None:
            v = GraphView.decide(self, now)
            if v is not None:
                self._view = v
        return v if v is not None else GraphView(self, None, None)


class WeightedCSRGraph:
    """CSR of A[target, source] = value_e with EXPLICIT per-edge values — the edge-weighted variants of the
    medium recipes: GCNConv(x, edge_index, edge_weight) (medium/models.py:55-62, gcn_norm with weights) and
    DIFFormer's gcn_conv (medium/difformer.py:63-79, value = edge_weight * d_in * d_out).  The caller computes
    the values (one elementwise expression); entries are ordered by (target, source) like sgf_csr_build's, the
    transpose by (source, target).  No gradient flows to the values (edge weights are data, not parameters).
    Same interface as CSRGraph towards ops.spmm."""

    blocked = False

    def __init__(self, edge_index: torch.Tensor, values: torch.Tensor, num_nodes: int):
        K.check(edge_index, values)
        n = int(num_nodes)
        self.n, self.nnz, self.device = n, int(edge_index.shape[1]), edge_index.device
        self._src, self._tgt = edge_index[0].contiguous(), edge_index[1].contiguous()
        self._values = values.detach().to(_F32).contiguous()
        self.rowptr, self.colind, self.val = self._sorted(self._tgt, self._src)
        self.long_segments = long_row_segments(self.rowptr, self.nnz)
        self._t, self.t_long_segments, self.symmetric = None, 0, False

    def _sorted(self, rows, cols):
        perm = torch.argsort(rows * self.n + cols, stable=True)
        counts = torch.bincount(rows, minlength=self.n)
        rowptr = torch.zeros(self.n + 1, dtype=torch.int64, device=self.device)
        torch.cumsum(counts, 0, out=rowptr[1:])
        return rowptr, cols[perm].to(torch.int32), self._values[perm]

    def transposed(self):
        if self._t is None:
            self._t = self._sorted(self._src, self._tgt)
            self.t_long_segments = long_row_segments(self._t[0], self.nnz)
        return self._t

    def plan(self, dtype, transposed=False):
        return None


def weighted_graph(edge_index: torch.Tensor, edge_weight: torch.Tensor, num_nodes: int, value_fn, tag: str):
    """Cached WeightedCSRGraph; `value_fn(edge_index, edge_weight, n)` -> (edge_index', values) builds the
    normalised values once per (edge_index, edge_weight) pair."""
    key_tag = (tag, edge_weight.data_ptr(), edge_weight._version, tuple(edge_weight.shape))

    def build(ei, n):
        ei2, vals = value_fn(ei, edge_weight, n)
        g = WeightedCSRGraph(ei2, vals, n)
        g.edge_weight_ref = edge_weight          # pins the key tensor, like graph_cache pins edge_index
        return g

    return graph_cache.get(edge_index, num_nodes, factory=build, tag=key_tag)


# SGF_SPMM_BLOCK = "rows_per_block,lds_rows" overrides the block shape (default: 128 rows, all 144 KiB of LDS)
def _block_shape(dtype):
    import os
    cap = K.lds_rows_max(dtype)
    env = os.environ.get("SGF_SPMM_BLOCK", "")
    if env:
        r, c = (int(t) for t in env.split(","))
        return r, min(c, cap)
    return 128, cap


class BlockedPlan:
    """Row-block plan of one CSR (sgf_spmm_plan): which neighbour rows each block of rows stages in
    LDS, and the entry codes / values re-ordered so that the LDS entries lead every row."""

    def __init__(self, rowptr, colind, val, n: int, dtype, rows_per_block=None, lds_rows=None):
        r, c = _block_shape(dtype)
        self.rows_per_block = int(rows_per_block or r)
        self.lds_rows = int(lds_rows or c)
        (self.ecode, self.eval, self.nlds, self.sh_ptr, self.sh_cols, st) = K.spmm_plan(
            rowptr, colind, val, n, self.rows_per_block, self.lds_rows, LONG_ROW)
        self.lds_entries, self.staged_rows, self.unique_pairs, self.nnz = st
        self.lds_fraction = self.lds_entries / max(self.nnz, 1)


# Tile plan parameters: at most TILE_CAP staged sources per block (whole 32-source chunks), a source is staged when at
# least TILE_MIN_COUNT of the block's entries reference it (a staged source costs one 512-byte row of X plus one
# 512-byte tile column per block, a gathered entry 520 bytes each time; measured at ogbn-products scale: 2 -> 2.16 ms,
# 3 -> 2.21 ms), blocks of at most TILE_MAX_ROWS rows (256-row blocks of 8 waves, one per CU: 2.75 ms; 64: 4.4 ms).
TILE_CAP = 512
TILE_MIN_COUNT = 2
TILE_MAX_ROWS = 128
TILE_SPARSE_DENSITY = 0.12


def _tile_params():
    import os
    env = os.environ.get("SGF_SPMM_TILE", "")          # "cap,min_count,max_rows" (experiments)
    if env:
        c, m, r = (int(t) for t in env.split(","))
        return c, m, r
    return TILE_CAP, TILE_MIN_COUNT, TILE_MAX_ROWS


class TilePlan:
    """Plan of sgf_spmm_tile for one CSR: row blocks, the sources each block stages, the dense tiles as matrix-core
    fragments (hi + lo bf16; packed: sparse groups as entries, see K.tile_pack) and the CSR of the entries left on the
    gather path.  keep_dense=True keeps the unpacked fragments as .tiles (tests)."""

    def __init__(self, rowptr, colind, val, n: int, blk_row: torch.Tensor, cap=None, min_count=None, keep_dense=False):
        c, m, _ = _tile_params()
        self.blk_row, self.nb = blk_row, int(blk_row.numel()) - 1
        self.block_rows = int((blk_row[1:] - blk_row[:-1]).max()) if self.nb > 0 else 1
        self.cap, self.min_count = int(cap or c), int(min_count or m)
        (self.sh_ptr, self.sh_cols, self.tile_ptr, self.tiles, self.rem_rowptr, self.rem_col, self.rem_val,
         st) = K.tile_plan(rowptr, colind, val, n, blk_row, self.cap, self.min_count, LONG_ROW)
        self.tile_entries, self.staged_rows, _, self.nnz, self.fragments, self.rem_entries = st[:6]
        self.tile_fraction = self.tile_entries / max(self.nnz, 1)
        # stored entries per tile cell: below ~2 % a tile column moves more bytes than the gathers it replaces
        self.tile_density = self.tile_entries / max(self.fragments * 512, 1)
        self.long_segments = long_row_segments(self.rem_rowptr, None)
        self.grp, self.pool, units = K.tile_pack(blk_row, self.tile_ptr, self.tiles, self.fragments)
        self.tile_bytes = units * 16                       # what a launch reads of them (dense: fragments * 2048)
        if not keep_dense:
            self.tiles = None
        self.bytes = (self.pool.numel() + self.grp.numel() * 4 + self.rem_col.numel() * 8 + self.rem_rowptr.numel() * 8
                      + self.sh_cols.numel() * 4)


# When to re-order (SGF_REORDER): "auto" (default) tries once per cached graph with at least
# REORDER_MIN_NODES nodes, at its second forward (a graph seen once is a mini-batch: planning would cost
# more than it saves) or when prepare_graph() asks for it; "1" tries every graph at first use; "0" never.
# The order is adopted only if the row-block plan on the re-ordered CSR then serves at least
# REORDER_MIN_LDS_FRACTION of the stored entries from LDS — a uniform random graph is an expander, no order
# helps it, and it keeps the plain kernel on its original CSR.
REORDER_MIN_NODES = 100_000
REORDER_MIN_LDS_FRACTION = 0.25
REORDER_ITERS = (6, 6)
_reorder_mode = None


def set_reorder_mode(mode: Optional[str]):
    """'auto' | 'always' | 'never' | None (= read SGF_REORDER).  Returns the previous setting."""
    global _reorder_mode
    prev, _reorder_mode = _reorder_mode, mode
    return prev


def _mode() -> str:
    if _reorder_mode is not None:
        return _reorder_mode
    import os
    return {"0": "never", "1": "always"}.get(os.environ.get("SGF_REORDER", "auto"), "auto")


class GraphView:
    """(graph, perm, inv): `graph` is the CSRGraph the layers multiply with; when perm is not None the
    model's rows are in the re-ordered numbering: row p of every activation is original node perm[p],
    and original node v sits at row inv[v]."""

    def __init__(self, graph, perm, inv, stats=None):
        self.graph, self.perm, self.inv, self.stats = graph, perm, inv, stats or {}

    @staticmethod
    def decide(g: "CSRGraph", now: bool):
        mode = _mode()
        if mode == "never" or K.name != "hip" or g.nnz == 0 or g.nnz >= 2 ** 32 - 1:
            return GraphView(g, None, None, {"reordered": False, "why": "disabled"})
        if mode == "auto":
            if g.n < REORDER_MIN_NODES:
                return GraphView(g, None, None, {"reordered": False, "why": "small graph"})
            if not now and g.forward_calls < 2:
                return None                    # undecided: wait for the second forward on this graph
        import os
        perm, inv, comm = K.reorder(g.edge_index, g.n, *REORDER_ITERS)
        g2 = CSRGraph(inv.long()[g.edge_index], g.n, validate=False)
        # how much neighbour sharing the new order exposes: the share of stored entries that fall into the dense
        # tiles of community-aligned row blocks — the adoption criterion, whichever kernel then runs
        _, _, max_rows = _tile_params()
        g2.blk_row = K.tile_blocks(comm[perm.long()].contiguous(), g.n, max_rows, g.device)
        try:
            tp = g2.tile_plan(False)
        except (ValueError, _lib.SgfError) as e:
            # a plan the tile kernel cannot take (a row block out of range, nnz beyond its 32-bit offsets): keep the graph
            # as given rather than fail the forward
            return GraphView(g, None, None, {"reordered": False, "why": f"tile plan unsupported: {e}"})
        if tp.tile_density < TILE_SPARSE_DENSITY and tp.min_count == TILE_MIN_COUNT and not os.environ.get("SGF_SPMM_TILE"):
            # tiles this sparse (a skewed graph: many sources referenced just twice per block) move more fragment bytes
            # than the gathers they replace: stage only sources with one more reference (power-law community graph at
            # ogbn-products scale: density 0.08 -> 0.11, 3.08 -> 2.92 ms; profiles/r03_spmm_tile.md)
            g2.tile_min_count = TILE_MIN_COUNT + 1
            g2._tile_plans.clear()
            tp = g2.tile_plan(False)
        stats = {"lds_fraction": tp.tile_fraction, "tile_density": tp.tile_density, "blocks": tp.nb,
                 "staged_rows_per_node": tp.staged_rows / max(g.n, 1), "plan_bytes": tp.bytes, "min_count": tp.min_count}
        if tp.tile_fraction < REORDER_MIN_LDS_FRACTION and mode != "always":
            return GraphView(g, None, None, {**stats, "reordered": False, "why": "no reuse to exploit"})
        # Which kernel multiplies with the re-ordered CSR: dense matrix-core tiles + gather remainder (sgf_spmm_tile)
        # for bf16 rows of 128 / 256 features; the flattened stream kernel (sgf_spmm_stream) otherwise and under
        # SGF_SPMM_TILED=0; SGF_SPMM_BLOCKED=1 selects r02's LDS-staged row blocks (profiles/r02_spmm_structured.md).
        g2.locality = True        # its gathers mostly hit in L2: ops.spmm_on picks sgf_spmm_stream
        g2.blocked = os.environ.get("SGF_SPMM_BLOCKED", "0") == "1"
        g2.tiled = os.environ.get("SGF_SPMM_TILED", "1") == "1" and not g2.blocked
        if not g2.tiled:
            g2._tile_plans.clear()
        stats["kernel"] = ("row-block (LDS-staged)" if g2.blocked else
                           "tiles (matrix cores) + gather remainder for bf16 rows of 128 / 256 features, else stream"
                           if g2.tiled else "stream")
        return GraphView(g2, perm, inv, {**stats, "reordered": True})


def prepare_graph(edge_index: torch.Tensor, num_nodes: int) -> GraphView:
    """Build everything that is per-graph, not per-step: the CSR, and (policy above) the node order and
    row-block plan.  The trainers need not call this — the same work happens lazily in the first two
    forwards — bench.py does, so that it stays outside the timed region like the CSR build."""
    return graph_cache.get(edge_index, num_nodes).view(now=True)


class _GraphCache:
    """edge_index -> CSRGraph, keyed on tensor identity AND version: the
    mini-batch trainers hand in a fresh edge_index every step, so entries are bounded (LRU) and each
    entry pins its key tensor so a recycled data_ptr can never alias a stale graph."""

    def __init__(self, capacity: int = 4):
        self.capacity = capacity
        self._d: "OrderedDict[tuple, CSRGraph]" = OrderedDict()

    def get(self, edge_index: torch.Tensor, num_nodes: int, factory=None, tag=None):
        key = (edge_index.data_ptr(), edge_index._version, tuple(edge_index.shape),
               tuple(edge_index.stride()), str(edge_index.device), int(num_nodes), tag)
        g = self._d.get(key)
        if g is not None:  # the entry pins its tensor, so an equal key means the same live memory
            self._d.move_to_end(key)
            return g
        g = (factory or CSRGraph)(edge_index, num_nodes)
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
def _sharded_spmm(graph, x, shard, transposed: bool):
    """Local rows of A (or of A^T) times the node-sharded operand.

    Halo path (graph.halo(...).enabled): pack the rows the peers need, one all_to_all_single, product on
    [own rows ; halo rows] with the block's relabelled columns — bytes on the links = the distinct cut-edge
    sources only.  Fallback when the cut is (nearly) everything: all-gather of the operand PIPELINED against
    the product — the operand is split into column chunks, every chunk's all-gather is issued up front
    (asynchronously, on the collective's own stream), and the SpMM of chunk c starts as soon as chunk c has
    arrived while chunks c+1.. are still on the xGMI links."""
    rowptr, colind, val = graph.transposed() if transposed else (graph.rowptr, graph.colind, graph.val)
    long_segments = graph.t_long_segments if transposed else graph.long_segments
    n_local = graph.n_local
    plan = graph.halo(shard, transposed) if hasattr(graph, "halo") else None
    if plan is not None and plan.enabled:
        if plan.n_halo == 0:                      # nothing to exchange (one rank, or a block without cut edges)
            return _own_block_spmm(graph, plan, (rowptr, plan.colind, val, long_segments), x, n_local, transposed)
        if getattr(shard, "overlap", False):
            shard.overlapped_exchanges = getattr(shard, "overlapped_exchanges", 0) + 1
            # the entries whose source this rank owns are multiplied while the halo rows are on the links; the halo
            # entries follow when they have arrived (their sum is rounded to the storage dtype before it is added: bf16
            # rows with cut edges carry two roundings more than the single-GPU product)
            own, (rp_h, ci_h, va_h, seg_h) = plan.split(rowptr, val, n_local)
            recv, work, keep = shard.halo_exchange_start(x, plan)
            y = _own_block_spmm(graph, plan, own, x, n_local, transposed)
            work.wait()
            y.add_(K.spmm(rp_h, ci_h, va_h, recv, n_local, long_segments=seg_h))
            del keep
            return y
        return K.spmm(rowptr, plan.colind, val, shard.halo_exchange(x, plan), n_local, long_segments=long_segments)
    d = x.shape[1]
    chunks = shard.gather_chunks(d)
    if chunks <= 1:
        return K.spmm(rowptr, colind, val, shard.all_gather_rows(x), n_local, long_segments=long_segments)
    w = d // chunks
    pending = [shard.all_gather_rows(x[:, c * w:(c + 1) * w].contiguous(), async_op=True) for c in range(chunks)]
    y = torch.empty((n_local, d), dtype=x.dtype, device=x.device)
    for c, (buf, work) in enumerate(pending):
        work.wait()                                   # the compute stream waits for THIS chunk only
        K.spmm(rowptr, colind, val, buf, n_local, out=y[:, c * w:(c + 1) * w], long_segments=long_segments)
    return y


def _own_block_spmm(graph, plan, own, x, n_local: int, transposed: bool):
    """The square block of a rank's rows x the rank's OWN columns (the part of a node-sharded product that needs no
    exchange).  When the partition follows sgf_reorder's order (dist.Repartition sets graph.locality / comm_local) this
    block has the structure the re-ordered single-GPU graph has, and takes the same kernels: the matrix-core tile kernel
    (sgf_spmm_tile, bf16 rows of 128 / 256 features; plan built once per block and direction) or the stream kernel; the
    plain row kernel otherwise."""
    rp, ci, va, segs = own
    if getattr(graph, "locality", False):
        import os
        if (hasattr(K, "tile_supported") and K.tile_supported(x.shape[1], x.dtype) and getattr(graph, "comm_local", None) is not None
                and os.environ.get("SGF_SPMM_TILED", "1") == "1" and n_local >= 256
                and x.shape[0] * max(x.stride(0), x.shape[1]) * 2 < 2 ** 32 - 2048):
            plans = plan.__dict__.setdefault("_own_tile_plans", {})
            tp = plans.get("plan")
            if tp is None:
                try:
                    _, _, max_rows = _tile_params()
                    blk_row = K.tile_blocks(graph.comm_local, n_local, max_rows, x.device)
                    tp = TilePlan(rp, ci, va, n_local, blk_row)
                    if tp.tile_fraction < REORDER_MIN_LDS_FRACTION:
                        tp = False                              # too little to put on the matrix cores: stream kernel
                except (ValueError, _lib.SgfError):
                    tp = False
                plans["plan"] = tp
            if tp:
                return K.spmm_tile(tp, x, n_local)
        return K.spmm(rp, ci, va, x, n_local, long_segments=segs, stream_hint=True)
    return K.spmm(rp, ci, va, x, n_local, long_segments=segs)


class _SpMM(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, graph, shard):
        K.check(x)
        ctx.graph, ctx.shard = graph, shard
        if shard is not None:
            # node-sharded: rows of A local, X rows gathered from all ranks (halo all-gather)
            return _sharded_spmm(graph, x, shard, False)
        return spmm_on(graph, x, False)

    @staticmethod
    def backward(ctx, gy):
        graph, shard = ctx.graph, ctx.shard
        if shard is not None:
            # dX_local = (A^T dY)[local rows]: local rows of the CSR of A^T times the sharded dY
            return _sharded_spmm(graph, gy.contiguous(), shard, True), None, None
        return spmm_on(graph, gy.contiguous(), True), None, None


def spmm_on(graph, x: torch.Tensor, transposed: bool, out=None) -> torch.Tensor:
    """A x or A^T x on one GPU: the LDS-staged row-block kernel when the graph carries a plan for this
    storage dtype (GraphView adopted a re-ordered CSR), else the wave-per-row kernel."""
    rp, ci, va = graph.transposed() if transposed else (graph.rowptr, graph.colind, graph.val)
    segs = graph.t_long_segments if transposed else graph.long_segments
    plan = graph.plan(x.dtype, transposed) if (getattr(graph, "blocked", False) and x.shape[1] <= 256) else None
    if plan is not None:
        return K.spmm_blocked(rp, plan, x, graph.n, out=out, long_segments=segs)
    if getattr(graph, "tiled", False):
        if (K.tile_supported(x.shape[1], x.dtype)
                and x.shape[0] * max(x.stride(0), x.shape[1]) * 2 < 2 ** 32 - 2048      # = sgf_spmm_tile's own bound
                and (out is None or (out.stride(0) % 8 == 0 and out.data_ptr() % 16 == 0))):
            return K.spmm_tile(graph.tile_plan(transposed), x, graph.n, out=out)
        if not K.tile_supported(x.shape[1], x.dtype) and getattr(graph, "_tile_plans", None):
            # this graph is being multiplied in a storage dtype / width the tile kernel does not take (fp32 runs): the plan
            # GraphView.decide built for its adoption statistics (about the size of the CSR) is dropped, not kept resident
            graph._tile_plans.clear()
    if getattr(graph, "locality", False):
        return K.spmm(rp, ci, va, x, graph.n, out=out, long_segments=segs, stream_hint=True)
    return K.spmm(rp, ci, va, x, graph.n, out=out, long_segments=segs)


def spmm(graph, x: torch.Tensor, shard=None) -> torch.Tensor:
    """Y = A X with A the cached normalised adjacency (torch_sparse.matmul(adj, x) in the reference)."""
    return _SpMM.apply(x, graph, shard)
