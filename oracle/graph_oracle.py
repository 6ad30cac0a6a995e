"""CPU restatements (numpy) of the graph-side planning kernels — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and the CPU kernel table of tests/cpu_kernels.py import this
module; sgformer_amd/ never does.  The reference (qitianwu/SGFormer) has none of these steps — its
SpMM is the third-party torch_sparse.matmul at large/ours.py:34 — so there is no reference arithmetic
to restate here; what is pinned is (a) that the planning kernels of libsgf are deterministic and equal
to this plain restatement entry for entry, and (b), in the tests, that a re-ordered / row-blocked SpMM
equals the plain one (the product the reference computes) on the same inputs.

  reorder(ei, n, iters1, iters2)             == sgf_reorder        (csrc/reorder.hip)
  spmm_plan(rowptr, colind, val, n, R, ...)  == sgf_spmm_plan      (csrc/spmm_plan.hip)
  graph_prologue(ei, n)                      == sgf_graph_prologue (csrc/prologue.hip; the trainer prologue
                                                large/main.py:75-79)
"""
from __future__ import annotations

import numpy as np

COUNT_MAX = 1023


def _propagate(src, dst, n, label, iters, exclude_self):
    """`iters` synchronous rounds: node t adopts the label carried by most of its in-edges (s -> t),
    ties to the smallest label; a node without (counted) in-edges keeps its label."""
    src = np.asarray(src, dtype=np.int64)
    dst = np.asarray(dst, dtype=np.int64)
    ok = (src >= 0) & (src < n) & (dst >= 0) & (dst < n)
    if exclude_self:
        ok &= src != dst
    src, dst = src[ok], dst[ok]
    for _ in range(iters):
        if src.size == 0:
            break
        key = dst * n + label[src]
        uk, cnt = np.unique(key, return_counts=True)
        node, lab = uk // n, uk % n
        order = np.lexsort((lab, -cnt, node))           # by node, then votes descending, then label ascending
        node_s = node[order]
        first = np.ones(node_s.size, dtype=bool)
        first[1:] = node_s[1:] != node_s[:-1]
        new = label.copy()
        new[node_s[first]] = lab[order][first]
        label = new
    return label


def reorder(edge_index, n, iters1=6, iters2=6):
    """perm, inv, community — two-level label propagation order (see csrc/reorder.hip)."""
    ei = np.asarray(edge_index, dtype=np.int64)
    src, dst = ei[0], ei[1]
    lab1 = _propagate(src, dst, n, np.arange(n, dtype=np.int64), iters1, exclude_self=False)
    used = np.zeros(n, dtype=np.int64)
    used[lab1] = 1
    rank_of_label = np.cumsum(used) - used
    cid = rank_of_label[lab1]
    ok = (src >= 0) & (src < n) & (dst >= 0) & (dst < n)
    lab2 = _propagate(cid[src[ok]], cid[dst[ok]], n, np.arange(n, dtype=np.int64), iters2, exclude_self=True)
    comm_order = np.lexsort((np.arange(n), lab2))       # communities by (level-2 label, id)
    comm_rank = np.empty(n, dtype=np.int64)
    comm_rank[comm_order] = np.arange(n)
    perm = np.lexsort((np.arange(n), comm_rank[cid]))   # nodes by (community rank, id)
    inv = np.empty(n, dtype=np.int64)
    inv[perm] = np.arange(n)
    return perm.astype(np.int32), inv.astype(np.int32), cid.astype(np.int32)


def spmm_plan(rowptr, colind, val, n, rows_per_block, lds_rows, long_len):
    """ecode, eval, nlds, sh_ptr, sh_cols, stats — see csrc/spmm_plan.hip."""
    rowptr = np.asarray(rowptr, dtype=np.int64)
    colind = np.asarray(colind, dtype=np.int64)
    val = np.asarray(val, dtype=np.float32)
    nnz = int(colind.size)
    nb = (n + rows_per_block - 1) // rows_per_block
    lens = np.diff(rowptr)
    rowid = np.repeat(np.arange(n, dtype=np.int64), lens)
    is_long = lens[rowid] > long_len
    blk = rowid // rows_per_block
    ecode = colind.astype(np.int64).copy()
    flag = np.zeros(nnz, dtype=bool)
    sh_ptr = np.zeros(nb + 1, dtype=np.int64)
    sh_cols_parts = []
    eligible = np.nonzero(~is_long)[0]
    key = blk[eligible] * (n + 1) + colind[eligible]
    uk, inverse, cnt = np.unique(key, return_inverse=True, return_counts=True)
    ub, us = uk // (n + 1), uk % (n + 1)
    cclip = np.minimum(cnt, COUNT_MAX)
    order = np.lexsort((us, -cclip, ub))                # per block: most-referenced first, then source id
    ub_s = ub[order]
    head = np.searchsorted(ub_s, np.arange(nb), side="left")
    rank = np.arange(order.size) - head[ub_s]
    ok = (rank < lds_rows) & (cclip[order] >= 2)
    slot_of_unique = np.full(uk.size, -1, dtype=np.int64)
    slot_of_unique[order[ok]] = rank[ok]
    nsh = np.bincount(ub_s[ok], minlength=nb)
    sh_ptr[1:] = np.cumsum(nsh)
    sh_cols = np.zeros(int(sh_ptr[-1]), dtype=np.int32)
    sh_cols[sh_ptr[ub_s[ok]] + rank[ok]] = us[order][ok]
    slots = slot_of_unique[inverse]
    hit = slots >= 0
    ecode[eligible[hit]] = (1 << 31) | slots[hit]
    flag[eligible[hit]] = True
    # stable partition inside each row: LDS entries first
    part = np.lexsort((np.arange(nnz), ~flag, rowid))
    ecode_out = ecode[part]
    eval_out = val[part]
    nlds = np.bincount(rowid[flag], minlength=n).astype(np.int32)
    ecode_out = np.where(ecode_out >= (1 << 31), ecode_out - (1 << 32), ecode_out).astype(np.int32)
    stats = np.array([int(flag.sum()), int(sh_ptr[-1]), int(uk.size) + int(is_long.any()), nnz], dtype=np.int64)
    return ecode_out, eval_out, nlds, sh_ptr.astype(np.int32), sh_cols, stats


def spmm_blocked(rowptr, ecode, eval_, nlds, sh_ptr, sh_cols, x, rows_per_block):
    """Y = A X evaluated THROUGH the plan (slot -> staged row -> source id), in float64."""
    rowptr = np.asarray(rowptr, dtype=np.int64)
    n = rowptr.size - 1
    x = np.asarray(x, dtype=np.float64)
    y = np.zeros((n, x.shape[1]), dtype=np.float64)
    for r in range(n):
        b = r // rows_per_block
        for e in range(rowptr[r], rowptr[r + 1]):
            c = int(ecode[e])
            if c < 0:
                c = int(sh_cols[sh_ptr[b] + (c & 0x7FFFFFFF)])
            y[r] += float(eval_[e]) * x[c]
    return y


def graph_prologue(edge_index, n, undirected=True):
    """The trainer prologue large/main.py:75-79 (PyG 1.7.2 semantics): to_undirected (symmetrise +
    coalesce = sort by (row, col), drop duplicates), remove_self_loops, add_self_loops (one (i, i) per
    node, appended).  Returns int64 [2, nnz']."""
    ei = np.asarray(edge_index, dtype=np.int64)
    src, dst = ei[0], ei[1]
    if undirected:
        src, dst = np.concatenate([src, dst]), np.concatenate([dst, src])
        key = np.unique(src * n + dst)
        src, dst = key // n, key % n
    keep = src != dst
    loops = np.arange(n, dtype=np.int64)
    return np.stack([np.concatenate([src[keep], loops]), np.concatenate([dst[keep], loops])])
