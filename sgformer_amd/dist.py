"""Node-sharded multi-GPU execution: one process per GPU, torch.distributed over RCCL/xGMI.

The reference has no distributed path at all; this is new.  Nodes are split
into contiguous, equal ranges — rank r owns rows [r*n_max, min((r+1)*n_max, N)) — parameters are
replicated, and the model runs on `model(x_local, edge_index_global)`.  The hot path has exactly
four exchange points, all designed into the kernels' partial-sum layouts:

  attention fwd/bwd   one all-reduce each of [K^T V | sum K | ||Q||^2 | ||K||^2] (H(d^2+d)+2 fp32)
                      and [dS0 | dz0 | .] (H(d^2+d)+1 fp32): <= 257 KB at d = 256 — latency bound
  SpMM fwd/bwd        HALO exchange of the operand rows (X forward, dY backward): at graph build every
                      rank lists the distinct remote columns its row block references, grouped by owner
                      (HaloPlan; the owners learn what to send by one all-to-all of index lists); per
                      SpMM each rank packs the rows its peers need (sgf_gather_rows), ONE
                      all_to_all_single moves them, and the product runs on [own rows ; halo rows] with
                      the block's columns relabelled once.  Bytes per SpMM = distinct cut-edge sources
                      x d x s instead of (P-1)/P x N x d x s.  When the halo would be more than
                      SGF_HALO_MAX (default 0.5) of the remote rows on any rank — a uniform random graph
                      cuts (P-1)/P of its edges, its halo IS the whole matrix — every rank falls back to
                      the all-gather of the operand in up to 4 column chunks whose gathers are pipelined
                      against the chunk SpMMs (ops._sharded_spmm), indexed by GLOBAL node id.  Directed
                      graphs: the backward multiplies with the row block of A^T, which carries its own
                      halo plan (the distinct remote TARGETS of local sources).
  BatchNorm1d         all-reduce of [sum | sumsq] (2 x d fp32, two passes) forward and of
                      [sum dz | sum dz*xhat] backward — required for parity with full-graph BN
  parameter grads     ONE flat all-reduce (SUM) per step: each rank's autograd already produces the
                      gradient of the GLOBAL loss w.r.t. its local rows, so parameter gradients are
                      plain sums over ranks
The loss is normalised by the GLOBAL number of training rows (`sharded_nll_loss`).

Two input conventions: by default every rank passes the GLOBAL edge_index (the graph is replicated,
the row block is cut out of the full CSR); with `ShardContext(N, local_edges=True)` a rank passes
only the edges whose target it owns (symmetric graphs; BASELINE.json config 5, where the global
edge list does not fit one GPU) — see ShardedGraph.
"""
from __future__ import annotations

import os
from typing import Iterable, Optional

import torch
import torch.distributed as dist

from . import ops


def _mix64(x: torch.Tensor, stream: int) -> torch.Tensor:
    """splitmix64 finaliser on int64 tensors (two's-complement wrap = arithmetic mod 2^64; the
    logical right shifts are emulated by masking the sign extension away)."""
    def shr(v, k):
        return (v >> k) & ((1 << (64 - k)) - 1)
    def c(u):   # unsigned 64-bit constant as the int64 with the same bits
        return u - (1 << 64) if u >= (1 << 63) else u
    z = x + c((0x9E3779B97F4A7C15 * (stream + 1)) & 0xFFFFFFFFFFFFFFFF)
    z = (z ^ shr(z, 30)) * c(0xBF58476D1CE4E5B9)
    z = (z ^ shr(z, 27)) * c(0x94D049BB133111EB)
    return z ^ shr(z, 31)


class HaloPlan:
    """Who needs which rows of a node-sharded operand, for one CSR row block with GLOBAL column ids.

    need     : sorted distinct remote columns of the block, i.e. grouped by owner rank (contiguous ranges)
    recv_counts[q] / send_counts[q] : rows this rank receives from / sends to rank q per exchange
    send_idx : LOCAL row numbers to pack, grouped by destination rank (what the peers asked for)
    colind   : the block's columns relabelled into [own rows (0..n_local) ; halo rows (n_local + position in need)]
    Built with three collectives (fraction all-reduce, counts all-to-all, index all-to-all), once per graph."""

    def __init__(self, colind: torch.Tensor, ctx: "ShardContext"):
        dev = colind.device
        cols = torch.unique(colind.long())                      # sorted
        remote = cols[(cols < ctx.r0) | (cols >= ctx.r1)]
        self.n_halo = int(remote.numel())
        n_remote_rows = max(ctx.n_global - ctx.n_local, 1)
        frac = torch.tensor([self.n_halo / n_remote_rows], dtype=torch.float64, device=dev)
        dist.all_reduce(frac, op=dist.ReduceOp.MAX, group=ctx.group)   # one decision for all ranks
        self.max_fraction = float(frac)
        self.enabled = self.max_fraction <= ctx.halo_max
        ctx.last_halo = (self.max_fraction, self.enabled)       # (what bench.py prints as `spmm_exchange`)
        if not self.enabled:
            return
        owner = torch.div(remote, ctx.n_max, rounding_mode="floor")
        need_counts = torch.bincount(owner, minlength=ctx.world)[: ctx.world]
        send_counts = torch.empty_like(need_counts)
        dist.all_to_all_single(send_counts, need_counts, group=ctx.group)
        self.recv_counts = [int(v) for v in need_counts.tolist()]
        self.send_counts = [int(v) for v in send_counts.tolist()]
        asked = torch.empty(sum(self.send_counts), dtype=torch.int64, device=dev)
        dist.all_to_all_single(asked, remote.contiguous(), self.send_counts, self.recv_counts, group=ctx.group)
        if asked.numel() and (int(asked.min()) < ctx.r0 or int(asked.max()) >= ctx.r1):
            raise RuntimeError("halo plan: a peer asked for a row this rank does not own")
        self.send_idx = (asked - ctx.r0).to(torch.int32)
        c = colind.long()
        is_local = (c >= ctx.r0) & (c < ctx.r1)
        pos = torch.searchsorted(remote, c.clamp(max=max(ctx.n_global - 1, 0)))
        self.colind = torch.where(is_local, c - ctx.r0, ctx.n_local + pos).to(torch.int32)
        self.n_ext = ctx.n_local + self.n_halo
        self._split = None

    def split(self, rowptr: torch.Tensor, val: torch.Tensor, n_local: int):
        """((rowptr, colind, val, long_segments) of the entries whose source this rank owns, the same for the entries
        whose source arrives with the halo — columns then index the RECEIVED rows): the own part can be multiplied
        while the exchange is on the links (ops._sharded_spmm).  Stored order inside every row is kept.  Built once."""
        if self._split is None:
            n_rows = int(rowptr.numel()) - 1
            c = self.colind.long()
            own = c < n_local
            rows = torch.repeat_interleave(torch.arange(n_rows, device=c.device), rowptr[1:] - rowptr[:-1])

            def part(mask, shift):
                rp = torch.zeros(n_rows + 1, dtype=torch.int64, device=c.device)
                torch.cumsum(torch.bincount(rows[mask], minlength=n_rows), 0, out=rp[1:])
                return rp, (c[mask] - shift).to(torch.int32).contiguous(), val[mask].contiguous(), ops.long_row_segments(rp, None)

            self._split = (part(own, 0), part(~own, n_local))
        return self._split


class ShardedGraph:
    """Local row block of the normalised adjacency (and of its transpose) with global column ids.

    Two ways in.  Default: every rank holds the GLOBAL edge_index; the block is cut out of the full
    CSR (bit-identical to the single-GPU arrays).  `ctx.local_edges`: the rank holds only the edges
    whose target it owns (graphs too large to replicate — the papers100M-shaped weak-scaling
    workload); the block is built from those alone plus one all-reduce of the degree vector.
    Column ids address the all-gathered operand, whose row g is global node g."""

    def __init__(self, edge_index: torch.Tensor, ctx: "ShardContext"):
        self.n = ctx.n_max * ctx.world          # rows of the gathered operand
        self.n_local = ctx.n_local
        if ctx.local_edges:
            self._init_from_local_edges(edge_index, ctx)
            return
        # Replicated edge list: only the edges this rank multiplies with are SORTED here (those whose target it
        # owns; for a directed graph also those whose source it owns, for the A^T block) — 1/P of the radix sort
        # of the full list; the in-degrees every value needs come from one O(E) histogram of all targets.
        n_g, dev = ctx.n_global, edge_index.device
        self.device = dev
        src, dst = edge_index[0], edge_index[1]
        # every rank holds the same list, so every rank raises (or none does) — an id outside [0, n) would otherwise
        # reach dinv[...] / colind as an out-of-bounds gather (the blocks below are built without validation)
        if edge_index.numel() > 0:
            lo, hi = torch.aminmax(edge_index)
            if int(lo) < 0 or int(hi) >= n_g:
                raise IndexError(f"edge_index has node ids outside [0, {n_g})")
        # in-degrees and the symmetry test: every rank works on ITS 1/P slice of the (replicated) edge list and one
        # all-reduce makes both global — O(E/P) per rank instead of an O(E) histogram + two 64-bit hashes everywhere
        e_all = int(src.numel())
        e0, e1 = ctx.rank * e_all // ctx.world, (ctx.rank + 1) * e_all // ctx.world
        deg = torch.bincount(dst[e0:e1], minlength=n_g)
        fwd, bwd = src[e0:e1] * n_g + dst[e0:e1], dst[e0:e1] * n_g + src[e0:e1]
        chk = torch.stack([_mix64(fwd, 0).sum(), _mix64(bwd, 0).sum(), _mix64(fwd, 1).sum(), _mix64(bwd, 1).sum()])
        del fwd, bwd
        ctx.all_reduce_exact(deg)
        ctx.all_reduce_exact(chk)                                     # sums wrap mod 2^64 on every rank alike
        deg = deg.to(torch.int32)
        self.symmetric = bool(chk[0] == chk[1]) and bool(chk[2] == chk[3])
        own_t = (dst >= ctx.r0) & (dst < ctx.r1)
        self.rowptr, self.colind, self.val = self._block(edge_index[:, own_t], deg, ctx, transposed=False)
        self.long_segments = ops.long_row_segments(self.rowptr)
        if self.symmetric:
            self._t = (self.rowptr, self.colind, self.val)
            self.t_long_segments = self.long_segments
        else:
            own_s = (src >= ctx.r0) & (src < ctx.r1)
            self._t = self._block(edge_index[:, own_s].flip(0), deg, ctx, transposed=True)
            self.t_long_segments = ops.long_row_segments(self._t[0])

    @staticmethod
    def _block(local_edges: torch.Tensor, deg: torch.Tensor, ctx: "ShardContext", transposed: bool):
        """CSR of the rows [r0, r1) from the edges that land in them (row = edge[1]), values from the GLOBAL
        in-degree vector with k_finalize's arithmetic (csr.hip): sqrt(1/d_target) * sqrt(1/d_source) in fp32,
        non-finite -> 0.  For the A^T block the rows are SOURCES, so (row, col) = (source, target)."""
        part = ops.CSRGraph(local_edges.contiguous(), ctx.n_global, validate=False)   # foreign rows come out empty
        lo = int(part.rowptr[ctx.r0])
        rowptr = (part.rowptr[ctx.r0:ctx.r1 + 1] - lo).contiguous()
        colind = part.colind
        dinv = (1.0 / deg.to(torch.float32)).sqrt()
        counts = rowptr[1:] - rowptr[:-1]
        row_of = torch.repeat_interleave(torch.arange(ctx.r0, ctx.r1, device=colind.device), counts,
                                         output_size=int(colind.numel()))
        tgt, srcn = (colind.long(), row_of) if transposed else (row_of, colind.long())
        val = dinv[tgt] * dinv[srcn]
        return rowptr, colind, torch.nan_to_num(val, nan=0.0, posinf=0.0, neginf=0.0)

    def _init_from_local_edges(self, edge_index: torch.Tensor, ctx: "ShardContext"):
        """edge_index = exactly the edges whose target (edge_index[1]) this rank owns, GLOBAL ids.
        sgf_csr_build sorts them and counts the in-degrees (large/ours.py:26-33: `d = degree(col, N)`
        — all edges into a node live on its owner, so the local count IS the global one for owned
        nodes and zero elsewhere); one all-reduce makes the degree vector global, from which the
        values 1/sqrt(d_target) * 1/sqrt(d_source) follow.  The transpose block needed by the backward
        (edges whose SOURCE is local) is the same block iff the global graph is symmetric: that is
        the contract of this mode, checked by an order-sensitive checksum over all ranks."""
        full = ops.CSRGraph(edge_index, ctx.n_global)      # foreign rows come out empty
        self.device = full.device
        lo, hi = int(full.rowptr[ctx.r0]), int(full.rowptr[ctx.r1])
        src, dst = edge_index[0], edge_index[1]
        # both contract checks ride on ONE collective so that every rank raises (or none does)
        # symmetry: the multiset {(s, t)} must equal {(t, s)}.  Sum of a NON-LINEAR per-edge hash (two
        # independent splitmix64 streams): a linear checksum such as sum(s*c + t) is blind to every
        # directed graph whose in- and out-degrees agree node by node (a directed ring passes it).
        n_g = ctx.n_global
        fwd, bwd = src * n_g + dst, dst * n_g + src
        chk = torch.stack([_mix64(fwd, 0).sum(), _mix64(bwd, 0).sum(),                     # wrap mod 2^64
                           torch.tensor(full.nnz - (hi - lo), dtype=torch.int64, device=self.device),
                           _mix64(fwd, 1).sum(), _mix64(bwd, 1).sum()])
        ctx.all_reduce_exact(chk)
        if int(chk[2]) != 0:
            raise ValueError(f"local_edges mode: {int(chk[2])} edge(s) were handed to a rank that does not own "
                             f"their target (rank {ctx.rank}: {full.nnz - (hi - lo)} outside [{ctx.r0}, {ctx.r1}))")
        if int(chk[0]) != int(chk[1]) or int(chk[3]) != int(chk[4]):
            raise ValueError("local_edges mode needs a symmetric global graph (every edge present in both "
                             "directions): the backward applies the same row block as A^T")
        deg = full.deg.clone()
        ctx.all_reduce_exact(deg)                            # global in-degree, int32 [N]
        self.rowptr = (full.rowptr[ctx.r0:ctx.r1 + 1] - lo).contiguous()
        self.colind = full.colind
        # same arithmetic as k_finalize (csr.hip): sqrt(1/d_t) * sqrt(1/d_s) in fp32, inf -> 0
        dinv = (1.0 / deg.to(torch.float32)).sqrt()
        counts = self.rowptr[1:] - self.rowptr[:-1]
        row_of = torch.repeat_interleave(torch.arange(ctx.r0, ctx.r1, device=self.device), counts, output_size=full.nnz)
        val = dinv[row_of] * dinv[self.colind.long()]
        self.val = torch.nan_to_num(val, nan=0.0, posinf=0.0, neginf=0.0)
        self.long_segments = ops.long_row_segments(self.rowptr)
        self.symmetric = True
        self._t = (self.rowptr, self.colind, self.val)
        self.t_long_segments = self.long_segments

    @staticmethod
    def _slice(rowptr, colind, val, ctx):
        lo, hi = int(rowptr[ctx.r0]), int(rowptr[ctx.r1])
        return ((rowptr[ctx.r0:ctx.r1 + 1] - lo).contiguous(), colind[lo:hi].clone(),
                val[lo:hi].clone())

    def transposed(self):
        return self._t

    def halo(self, ctx: "ShardContext", transposed: bool):
        """HaloPlan of the forward block (or of the A^T block the backward multiplies with); the same plan
        when A is symmetric.  Built on first use — by every rank at the same point of the step."""
        if not hasattr(self, "_halo"):
            self._halo = {}
        key = bool(transposed) and not self.symmetric
        if key not in self._halo:
            colind = self._t[1] if key else self.colind
            self._halo[key] = HaloPlan(colind, ctx)
        return self._halo[key]


class RowExchange:
    """Moves node rows between two contiguous partitions of the same P ranks: rank r owns `want` = the (global,
    old-numbering) rows it needs, in the order it wants them.  One plan serves both directions: forward() gathers the
    wanted rows from their owners, backward() returns rows (or their gradients) to the owners.  Built with two
    collectives; every exchange is one pack (sgf_gather_rows) + one all_to_all_single + one placement gather."""

    def __init__(self, want: torch.Tensor, ctx: "ShardContext"):
        dev = want.device
        want = want.long()
        owner = torch.div(want, ctx.n_max, rounding_mode="floor")
        order = torch.argsort(owner, stable=True)                 # receive buffer position j  <-  wanted row order[j]
        self.place = torch.empty_like(order)
        self.place[order] = torch.arange(order.numel(), device=dev)   # wanted row i sits at recv position place[i]
        self.order = order
        need_counts = torch.bincount(owner, minlength=ctx.world)[: ctx.world]
        send_counts = torch.empty_like(need_counts)
        dist.all_to_all_single(send_counts, need_counts, group=ctx.group)
        self.recv_counts = [int(v) for v in need_counts.tolist()]
        self.send_counts = [int(v) for v in send_counts.tolist()]
        asked = torch.empty(sum(self.send_counts), dtype=torch.int64, device=dev)
        dist.all_to_all_single(asked, want[order].contiguous(), self.send_counts, self.recv_counts, group=ctx.group)
        if asked.numel() and (int(asked.min()) < ctx.r0 or int(asked.max()) >= ctx.r1):
            raise RuntimeError("row exchange: a peer asked for a row this rank does not own")
        self.send_idx = asked - ctx.r0                           # local rows to pack, grouped by destination
        self.n_want, self.n_own, self.ctx = int(want.numel()), ctx.n_local, ctx
        # when every owned row is wanted exactly once (`want` = a permutation of all nodes: Repartition), the way back is
        # a GATHER by the inverse of send_idx, not a scatter into zeros
        self.send_inv = None
        if int(self.send_idx.numel()) == self.n_own and self.n_own > 0:
            inv = torch.full((self.n_own,), -1, dtype=torch.int64, device=dev)
            inv[self.send_idx] = torch.arange(self.n_own, device=dev)
            if int(inv.min()) >= 0:
                self.send_inv = inv

    @staticmethod
    def _rows(t: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
        """t[idx] — on sgf_gather_rows for [n, d] float matrices (16 bytes per lane; ATen's advanced indexing launches
        an index kernel with one thread per element and materialises the result through a second copy)."""
        if t.dim() == 2 and t.dtype in (torch.float32, torch.bfloat16) and t.shape[0] > 0 and idx.numel() > 0:
            return ops.gather_rows(t, idx)
        return t[idx].contiguous()

    def forward(self, t: torch.Tensor) -> torch.Tensor:
        """rows of the OLD partition (this rank's [n_local, ...]) -> the rows this rank wants, in `want` order."""
        send = self._rows(t, self.send_idx)
        recv = t.new_empty((self.n_want,) + tuple(t.shape[1:]))
        dist.all_to_all_single(recv, send, self.recv_counts, self.send_counts, group=self.ctx.group)
        self.ctx.bytes_repartition += send.numel() * send.element_size()
        return self._rows(recv, self.place)

    def backward(self, t: torch.Tensor) -> torch.Tensor:
        """the reverse move: rows in `want` order -> back to their owners' local positions (every owned row is wanted
        by exactly one rank when `want` is a permutation of all nodes)."""
        send = self._rows(t, self.order)
        recv = t.new_empty((int(self.send_idx.numel()),) + tuple(t.shape[1:]))
        dist.all_to_all_single(recv, send, self.send_counts, self.recv_counts, group=self.ctx.group)
        self.ctx.bytes_repartition += send.numel() * send.element_size()
        if self.send_inv is not None:
            return self._rows(recv, self.send_inv)
        out = t.new_zeros((self.n_own,) + tuple(t.shape[1:]))
        out[self.send_idx] = recv
        return out


class _Exchange(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t, plan, reverse: bool):
        ctx.plan, ctx.reverse = plan, reverse
        return plan.backward(t) if reverse else plan.forward(t)

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        return (ctx.plan.forward(g) if ctx.reverse else ctx.plan.backward(g)), None, None


class Repartition:
    """A locality-restoring node order ACROSS ranks (the halo exchange only engaged when the caller's
    ids already carried the locality).  Every rank holds the global edge list (replicated mode); rank 0 computes the
    sgf_reorder permutation on it and broadcasts it, every rank relabels the edges and takes the contiguous
    range [r0, r1) of the NEW numbering; features enter and logits leave through one all-to-all each
    (`to_new` / `to_old`, differentiable), so the caller keeps its own numbering.  Adopted only if the halo of the
    re-partitioned graph is small enough for the halo path (HaloPlan.enabled) — a uniform random graph keeps its
    order and the all-gather fallback."""

    def __init__(self, edge_index: torch.Tensor, ctx: "ShardContext"):
        n, dev = ctx.n_global, edge_index.device
        # ONE rank runs sgf_reorder (label propagation over the whole edge list: O(E) sorts); the others receive the
        # order and the community labels — 8 N bytes on the links instead of P - 1 redundant runs
        po = torch.empty((2, n), dtype=torch.int32, device=dev)
        if ctx.rank == 0:
            perm, _, comm = ops.K.reorder(edge_index, n, *ops.REORDER_ITERS)
            po[0], po[1] = perm.to(dev), comm.to(dev)[perm.long()]          # community of NEW row p
        src = 0 if ctx.group is None else dist.get_global_rank(ctx.group, 0)
        dist.broadcast(po, src=src, group=ctx.group)
        perm, comm_new = po[0].contiguous(), po[1].contiguous()
        inv = torch.empty(n, dtype=torch.int32, device=dev)
        inv[perm.long()] = torch.arange(n, dtype=torch.int32, device=dev)
        # would the re-partitioned graph take the halo path?  Decide from the halo FRACTION alone — the distinct remote
        # sources of the edges whose (new) target this rank owns — before any CSR is built: a uniform random graph keeps
        # its order (and the all-gather), and nothing is constructed just to be thrown away
        new_src, new_dst = inv[edge_index[0]].long(), inv[edge_index[1]].long()
        mine = (new_dst >= ctx.r0) & (new_dst < ctx.r1)
        cols = torch.unique(new_src[mine])
        n_halo = int(((cols < ctx.r0) | (cols >= ctx.r1)).sum())
        frac = torch.tensor([n_halo / max(ctx.n_global - ctx.n_local, 1)], dtype=torch.float64, device=dev)
        dist.all_reduce(frac, op=dist.ReduceOp.MAX, group=ctx.group)
        self.adopted = float(frac) <= ctx.halo_max
        self.stats = {"halo_fraction": float(frac), "adopted": self.adopted}
        if not self.adopted:
            return
        ei2 = torch.stack([new_src, new_dst])
        del new_src, new_dst, mine, cols
        ei2._sgf_trusted = True
        graph = ShardedGraph(ei2, ctx)
        # the rows of this rank follow the communities: its own-column block has the locality the re-ordered single-GPU
        # graph has — stream / tile kernels instead of the plain row kernel (ops._own_block_spmm)
        graph.locality = True
        graph.comm_local = comm_new[ctx.r0:ctx.r1].contiguous()
        self.edge_index, self.graph = ei2, graph
        ei2._sgf_sharded_graph = graph
        self.perm, self.inv = perm, inv
        self.plan = RowExchange(perm[ctx.r0:ctx.r1], ctx)        # new row p of this rank = old node perm[p]
        self._cache = None

    def to_new(self, t: torch.Tensor) -> torch.Tensor:
        """[n_local, ...] in the caller's partition -> this rank's rows of the re-ordered partition.  A tensor that
        does not require grad and is passed again (full-graph training: the same x every step) is moved once."""
        if not t.requires_grad:
            key = (t.data_ptr(), t._version, tuple(t.shape), t.dtype)
            if self._cache is not None and self._cache[0] == key:
                return self._cache[1]
            out = self.plan.forward(t)
            self._cache = (key, out, t)
            return out
        return _Exchange.apply(t, self.plan, False)

    def to_old(self, t: torch.Tensor) -> torch.Tensor:
        return _Exchange.apply(t, self.plan, True)


class ShardContext:
    """Partition + collectives of one rank.  `group=None` uses the default process group."""

    def __init__(self, n_global: int, group=None, local_edges: bool = False):
        """local_edges=True: `model(x_local, edge_index)` receives only the edges whose target this
        rank owns (ShardedGraph._init_from_local_edges) instead of the global edge list."""
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.n_global = int(n_global)
        self.n_max = -(-self.n_global // self.world)
        self.r0 = min(self.rank * self.n_max, self.n_global)
        self.r1 = min(self.r0 + self.n_max, self.n_global)
        self.n_local = self.r1 - self.r0
        self.local_edges = bool(local_edges)
        self.halo_max = float(os.environ.get("SGF_HALO_MAX", "0.5"))   # largest halo / remote-rows ratio served by the halo path
        # own-column entries are multiplied while the halo rows are on the links (SGF_DIST_OVERLAP=0: one product on
        # [own rows ; halo rows] after the exchange, the r02 form)
        self.overlap = os.environ.get("SGF_DIST_OVERLAP", "1") != "0"
        self.bytes_all_reduced = 0
        self.bytes_all_gathered = 0
        self.bytes_halo_sent = 0
        self.bytes_repartition = 0
        self.last_halo = None           # (largest halo / remote-rows ratio over the ranks, halo path taken) of the last plan
        # SGF_DIST_REORDER=0 keeps the caller's node order across ranks (default: try sgf_reorder once per graph)
        self.reorder = os.environ.get("SGF_DIST_REORDER", "1") == "1" and not self.local_edges
        self.local_graph = False        # batch mode: the SpMM multiplies with this rank's own edges only (no halo)

    @classmethod
    def for_batch(cls, n_local_rows: int, group=None) -> "ShardContext":
        """BASELINE.json config 5, 'mini-batch + 8-GPU node-shard' (SURVEY.md §8e): every rank draws a mini-batch from
        ITS node shard; the union of the ranks' batches is the attention set (the N of large/ours.py:133-148 is the
        GLOBAL batch size, the K^T V / BatchNorm partials are all-reduced), while the GCN branch multiplies with the
        rank's own induced subgraph — edges inside the local batch only, no halo.  Batches may differ in size: ranges
        follow an all-gather of the local counts.  Parity is defined against ONE process running the concatenated
        batch with the block-diagonal union of the ranks' subgraphs (tests/test_dist.py)."""
        ctx = cls.__new__(cls)
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        ctx.group = group
        ctx.rank, ctx.world = dist.get_rank(group), dist.get_world_size(group)
        counts = [None] * ctx.world
        dist.all_gather_object(counts, int(n_local_rows), group=group)
        ctx.n_global = int(sum(counts))
        ctx.n_max = max(max(counts), 1)
        ctx.r0 = int(sum(counts[: ctx.rank]))
        ctx.r1 = ctx.r0 + int(n_local_rows)
        ctx.n_local = int(n_local_rows)
        ctx.local_edges, ctx.local_graph, ctx.reorder = False, True, False
        ctx.halo_max = 0.0
        ctx.bytes_all_reduced = ctx.bytes_all_gathered = ctx.bytes_halo_sent = ctx.bytes_repartition = 0
        ctx.last_halo = None
        return ctx

    def repartition_for(self, edge_index: torch.Tensor):
        """The Repartition of this (replicated) edge list, or None when disabled / not adopted; decided once per
        edge_index, at the same point of the step on every rank."""
        if not self.reorder or not hasattr(ops.K, "reorder"):
            return None
        cache = self.__dict__.setdefault("_repart", {})
        key = (edge_index.data_ptr(), edge_index._version, tuple(edge_index.shape))
        if key not in cache:
            cache.clear()
            rp = Repartition(edge_index, self)
            cache[key] = (rp if rp.adopted else None, edge_index)
        return cache[key][0]

    # ---- partition helpers ----
    def shard_rows(self, t: torch.Tensor) -> torch.Tensor:
        return t[self.r0:self.r1]

    def local_index(self, global_idx: torch.Tensor) -> torch.Tensor:
        m = (global_idx >= self.r0) & (global_idx < self.r1)
        return global_idx[m] - self.r0

    def graph_for(self, edge_index: torch.Tensor) -> ShardedGraph:
        g = getattr(edge_index, "_sgf_sharded_graph", None)       # the relabelled list of a Repartition carries its graph
        if g is not None:
            return g
        return ops.graph_cache.get(edge_index, -self.n_global - self.rank - 1,
                                   factory=lambda ei, _n: ShardedGraph(ei, self))

    # ---- collectives ----
    def all_reduce(self, t: torch.Tensor) -> torch.Tensor:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        self.bytes_all_reduced += t.numel() * t.element_size()
        return t

    def all_reduce_exact(self, t: torch.Tensor) -> torch.Tensor:
        """Integer SUM all-reduce (graph construction: degrees, checksums); not counted as step traffic."""
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t

    def all_gather_rows(self, x: torch.Tensor, async_op: bool = False):
        """[n_local, d] on every rank -> [world * n_max, d]; row g is global node g (rows >= N pad).
        async_op=True returns (buffer, work): the buffer is valid after work.wait()."""
        d = x.shape[1]
        if x.shape[0] != self.n_max:
            pad = torch.zeros((self.n_max, d), dtype=x.dtype, device=x.device)
            pad[: x.shape[0]] = x
            x = pad
        out = torch.empty((self.world * self.n_max, d), dtype=x.dtype, device=x.device)
        work = dist.all_gather_into_tensor(out, x.contiguous(), group=self.group, async_op=async_op)
        self.bytes_all_gathered += out.numel() * out.element_size()
        return (out, work) if async_op else out

    def halo_exchange(self, x: torch.Tensor, plan: "HaloPlan") -> torch.Tensor:
        """[n_local, d] -> [n_local + n_halo, d]: own rows, then the rows of the peers this rank's block
        references (in `plan.need` order).  One pack kernel + one all_to_all_single."""
        d = x.shape[1]
        send = ops.gather_rows(x, plan.send_idx) if plan.send_idx.numel() else x.new_empty((0, d))
        ext = torch.empty((plan.n_ext, d), dtype=x.dtype, device=x.device)
        ext[: self.n_local] = x
        dist.all_to_all_single(ext[self.n_local:], send, plan.recv_counts, plan.send_counts, group=self.group)
        self.bytes_halo_sent += send.numel() * send.element_size()
        return ext

    def halo_exchange_start(self, x: torch.Tensor, plan: "HaloPlan"):
        """(rows of the peers in `plan.need` order [n_halo, d], work): the exchange is issued asynchronously (on the
        collective's own stream); the buffer is valid after work.wait()."""
        d = x.shape[1]
        send = ops.gather_rows(x, plan.send_idx) if plan.send_idx.numel() else x.new_empty((0, d))
        recv = torch.empty((plan.n_halo, d), dtype=x.dtype, device=x.device)
        work = dist.all_to_all_single(recv, send, plan.recv_counts, plan.send_counts, group=self.group, async_op=True)
        self.bytes_halo_sent += send.numel() * send.element_size()
        return recv, work, send

    def gather_chunks(self, d: int) -> int:
        """Column chunks the SpMM operand is gathered in (ops._sharded_spmm): up to 4, each at least
        SGF_DIST_CHUNK_COLS (default 64) columns wide and a multiple of 4; 1 = no pipelining."""
        width = max(4, int(os.environ.get("SGF_DIST_CHUNK_COLS", "64")))
        c = max(1, min(4, d // width))
        while c > 1 and d % (4 * c) != 0:
            c -= 1
        return c

    def unsum(self, t: Optional[torch.Tensor]):
        return None if t is None else t / self.world

    def sync_grads(self, params: Iterable[torch.nn.Parameter]):
        """One flat all-reduce (SUM) over every parameter gradient."""
        grads = [p.grad for p in params if p.grad is not None]
        if not grads:
            return
        flat = torch.cat([g.reshape(-1) for g in grads])
        self.all_reduce(flat)
        off = 0
        for g in grads:
            n = g.numel()
            g.copy_(flat[off:off + n].view_as(g))
            off += n


def shard_model(model: torch.nn.Module, ctx: Optional[ShardContext]):
    """Point every exchange-carrying submodule of a drop-in SGFormer at `ctx` (None = unshard)."""
    for m in model.modules():
        if hasattr(m, "_shard"):
            m._shard = ctx
    return model


def sharded_nll_loss(logits_local: torch.Tensor, y_local: torch.Tensor, train_idx_local: torch.Tensor,
                     n_train_global: int) -> torch.Tensor:
    """log_softmax + NLL (large/main.py:139-141) summed over the LOCAL training rows and divided by
    the GLOBAL count: the sum over ranks is the full-graph mean loss (fused: ops.nll_loss_rows)."""
    return ops.nll_loss_rows(logits_local, y_local, train_idx_local, denom=n_train_global)
