"""CPU restatements (numpy) of the graph-side planning kernels — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and the CPU kernel table of tests/cpu_kernels.py import this
module; sgformer_amd/ never does.  The reference (qitianwu/SGFormer) has none of these steps — its
SpMM is the third-party torch_sparse.matmul at large/ours.py:34 — so there is no reference arithmetic
to restate here; what is pinned is (a) that the planning kernels of libsgf are deterministic and equal
to this plain restatement entry for entry, and (b), in the tests, that a re-ordered / row-blocked SpMM
equals the plain one (the product the reference computes) on the same inputs.

  reorder(ei, n, iters1, iters2)             == sgf_reorder        (csrc/reorder.hip)
  spmm_plan(rowptr, colind, val, n, R, ...)  == sgf_spmm_plan      (csrc/spmm_plan.hip)
  tile_blocks / tile_plan                    == sgf_spmm_tile_blocks / _plan / _fill (csrc/spmm_plan.hip)
  neighbor_sample(rowptr, colind, seeds, ...) == sgf_neighbor_sample_* (csrc/sampler.hip; NeighborLoader semantics of
                                                100M/nb-sample.py:125-151, this library's own random stream)
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


def _plan_core(rowptr, colind, val, n, blk_of_row, nb, lds_rows, long_len, min_count=2, slots_by_id=False):
    """Shared by spmm_plan and tile_plan: per block the `lds_rows` most-referenced sources with at least `min_count`
    references (most-referenced first, ties to the smaller id) get slots; entries towards them get the code
    0x80000000 | slot and move to the front of their row (stable).  Returns ecode, eval, nlds, nsh, cols (list of
    per-block source arrays), flags-sum, unique count."""
    rowptr = np.asarray(rowptr, dtype=np.int64)
    colind = np.asarray(colind, dtype=np.int64)
    val = np.asarray(val, dtype=np.float32)
    nnz = int(colind.size)
    lens = np.diff(rowptr)
    rowid = np.repeat(np.arange(n, dtype=np.int64), lens)
    is_long = lens[rowid] > long_len
    blk = np.asarray(blk_of_row, dtype=np.int64)[rowid]
    ecode = colind.astype(np.int64).copy()
    flag = np.zeros(nnz, dtype=bool)
    eligible = np.nonzero(~is_long)[0]
    key = blk[eligible] * (n + 1) + colind[eligible]
    uk, inverse, cnt = np.unique(key, return_inverse=True, return_counts=True)
    ub, us = uk // (n + 1), uk % (n + 1)
    cclip = np.minimum(cnt, COUNT_MAX)
    order = np.lexsort((us, -cclip, ub))                # per block: most-referenced first, then source id
    ub_s = ub[order]
    head = np.searchsorted(ub_s, np.arange(nb), side="left")
    rank = np.arange(order.size) - head[ub_s]
    ok = (rank < lds_rows) & (cclip[order] >= min_count)
    slot_of_unique = np.full(uk.size, -1, dtype=np.int64)
    slot_of_unique[order[ok]] = rank[ok]
    if slots_by_id:
        # tile plans: the SELECTION is by count, the slots are dealt in ascending source id (uniques are sorted by
        # (block, source)): sibling blocks of one community then stage the same sources in the same order
        sel = slot_of_unique >= 0
        pos = np.cumsum(sel) - sel
        first_u = np.searchsorted(ub, np.arange(nb), side="left")
        first = np.append(pos, pos[-1] + sel[-1] if pos.size else 0)[np.minimum(first_u, uk.size)]
        slot_of_unique = np.where(sel, pos - first[ub], -1)
        rank_s = slot_of_unique[order]
        rank = np.where(ok, rank_s, rank)
    nsh = np.bincount(ub_s[ok], minlength=nb).astype(np.int64)
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
    staged = (ub_s[ok], rank[ok], us[order][ok])        # (block, slot, source) of every staged source
    return ecode_out, eval_out, nlds, nsh, staged, int(flag.sum()), int(uk.size) + int(is_long.any())


def spmm_plan(rowptr, colind, val, n, rows_per_block, lds_rows, long_len):
    """ecode, eval, nlds, sh_ptr, sh_cols, stats — see csrc/spmm_plan.hip."""
    nb = (n + rows_per_block - 1) // rows_per_block
    nnz = int(np.asarray(colind).size)
    ecode, ev, nlds, nsh, (sb, sr, ss), nflag, nuniq = _plan_core(
        rowptr, colind, val, n, np.arange(n, dtype=np.int64) // rows_per_block, nb, lds_rows, long_len)
    sh_ptr = np.zeros(nb + 1, dtype=np.int64)
    sh_ptr[1:] = np.cumsum(nsh)
    sh_cols = np.zeros(int(sh_ptr[-1]), dtype=np.int32)
    sh_cols[sh_ptr[sb] + sr] = ss
    stats = np.array([nflag, int(sh_ptr[-1]), nuniq, nnz], dtype=np.int64)
    return ecode, ev, nlds, sh_ptr.astype(np.int32), sh_cols, stats


def f32_to_bf16_bits(f):
    """round-to-nearest-even bf16 bit patterns of a float32 array (NaN preserved) — csrc/common.h f32_to_bf16."""
    u = np.asarray(f, dtype=np.float32).view(np.uint32).astype(np.uint64)
    nan = (u & 0x7FFFFFFF) > 0x7F800000
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) & 0xFFFF
    r = np.where(nan, (u >> 16) | 0x40, r)
    return r.astype(np.uint16)


def bf16_bits_to_f32(h):
    return (np.asarray(h, dtype=np.uint16).astype(np.uint32) << 16).view(np.float32)


def tile_blocks(comm_sorted, n, max_rows):
    """blk_row — sgf_spmm_tile_blocks (csrc/spmm_plan.hip): runs of one community packed greedily into blocks."""
    out = [0]
    if comm_sorted is None:
        out += list(range(max_rows, n, max_rows))
        if n > 0:
            out.append(n)
        return np.asarray(out, dtype=np.int32)
    comm = np.asarray(comm_sorted)
    cur, i = 0, 0
    while i < n:
        j = i + 1
        while j < n and comm[j] == comm[i]:
            j += 1
        ln = j - i
        if ln > max_rows:
            if cur > 0:
                out.append(i)
                cur = 0
            k = (ln + max_rows - 1) // max_rows
            piece = min(max_rows, ((ln + k - 1) // k + 31) // 32 * 32)
            pos = i
            while pos < j:
                pos = min(pos + piece, j)
                out.append(pos)
        else:
            if cur + ln > max_rows:
                out.append(i)
                cur = 0
            cur += ln
        i = j
    if cur > 0:
        out.append(n)
    return np.asarray(out, dtype=np.int32)


def tile_plan(rowptr, colind, val, n, blk_row, cap, min_count, long_len):
    """sh_ptr, sh_cols, tile_ptr, tiles (uint16 [fragments * 1024]), rem_rowptr, rem_col, rem_val, stats —
    sgf_spmm_tile_plan + sgf_spmm_tile_fill (csrc/spmm_plan.hip)."""
    rowptr = np.asarray(rowptr, dtype=np.int64)
    blk_row = np.asarray(blk_row, dtype=np.int64)
    nb = blk_row.size - 1
    nnz = int(np.asarray(colind).size)
    row_block = np.repeat(np.arange(nb, dtype=np.int64), np.diff(blk_row))
    ecode, ev, nlds, nsh, (sb, sr, ss), nflag, nuniq = _plan_core(rowptr, colind, val, n, row_block, nb, cap, long_len,
                                                                 min_count, slots_by_id=True)
    nshp = (nsh + 31) // 32 * 32
    sh_ptr = np.zeros(nb + 1, dtype=np.int64)
    sh_ptr[1:] = np.cumsum(nshp)
    sh_cols = np.repeat(blk_row[:-1], nshp).astype(np.int32)           # padding: the block's first row
    sh_cols[sh_ptr[sb] + sr] = ss
    rt = (np.diff(blk_row) + 31) // 32
    tile_ptr = np.zeros(nb + 1, dtype=np.int64)
    tile_ptr[1:] = np.cumsum(rt * (nshp // 16))
    nfrag = int(tile_ptr[-1])
    stage = np.zeros(nfrag * 512, dtype=np.float32)
    lens = np.diff(rowptr)
    rowid = np.repeat(np.arange(n, dtype=np.int64), lens)
    k_in_row = np.arange(nnz, dtype=np.int64) - rowptr[rowid]
    in_tile = k_in_row < nlds[rowid]
    r = rowid[in_tile]
    c = (ecode[in_tile].astype(np.int64)) & 0x7FFFFFFF
    b = row_block[r]
    m = r - blk_row[b]
    f = tile_ptr[b] + ((c >> 5) * rt[b] + (m >> 5)) * 2 + ((c >> 4) & 1)
    kk = c & 15
    lane = (m & 31) + 32 * (kk >> 3)
    np.add.at(stage, f * 512 + lane * 8 + (kk & 7), ev[in_tile])       # sequential: duplicates add in stored order
    hi = f32_to_bf16_bits(stage)
    lo = f32_to_bf16_bits(stage - bf16_bits_to_f32(hi))
    tiles = np.empty((nfrag, 2, 64, 8), dtype=np.uint16)
    tiles[:, 0] = hi.reshape(nfrag, 64, 8)
    tiles[:, 1] = lo.reshape(nfrag, 64, 8)
    # a row's gathered share is padded to an even count: the padding repeats the row's last source with value 0
    rlen = lens - nlds
    rpad = (rlen + 1) // 2 * 2
    rem_rowptr = np.zeros(n + 1, dtype=np.int64)
    rem_rowptr[1:] = np.cumsum(rpad)
    rem_col = np.zeros(int(rem_rowptr[-1]), dtype=np.int32)
    rem_val = np.zeros(int(rem_rowptr[-1]), dtype=np.float32)
    dst = rem_rowptr[rowid[~in_tile]] + (k_in_row[~in_tile] - nlds[rowid[~in_tile]])
    rem_col[dst] = ecode[~in_tile]
    rem_val[dst] = ev[~in_tile]
    odd = np.nonzero(rlen % 2 == 1)[0]
    rem_col[rem_rowptr[odd + 1] - 1] = ecode[rowptr[odd + 1] - 1]
    stats = np.array([nflag, int(sh_ptr[-1]), nuniq, nnz, nfrag, int(rem_rowptr[-1]), 0, 0], dtype=np.int64)
    return (sh_ptr.astype(np.int32), sh_cols, tile_ptr, tiles.reshape(-1), rem_rowptr, rem_col, rem_val, stats)


TILE_SPARSE_MAX = 384      # csrc/spmm_pack.hip kSparseMax


def tile_pack(blk_row, tile_ptr, tiles):
    """(grp int32 [groups, 2], pool uint8 [units * 16]) — the fragments of tile_plan in the form the kernel streams them
    (restates csrc/spmm_pack.hip; include/sgf.h, sgf_spmm_tile_pack*).  Group g = tile_ptr[b] / 2 + t * Q_b + q holds the
    two fragments (k-steps) of (block b, row tile t, chunk q), stored by the plan at tile_ptr[b] + (q * T_b + t) * 2.
    <= TILE_SPARSE_MAX occupied cells (hi or lo != 0): the cells in image order ([k-step][lane][j]) as entries
    {uint32 byte offset of the hi element in the 4 KiB [k-step][hi, lo][lane][8] image, uint32 hi | lo << 16}, padded
    with a zero entry to an even count; otherwise the 4 KiB image, grp[g, 1] = -1.  grp[g, 0] = offset / 16."""
    blk_row = np.asarray(blk_row, dtype=np.int64)
    tile_ptr = np.asarray(tile_ptr, dtype=np.int64)
    t = np.asarray(tiles, dtype=np.uint16).reshape(-1, 2, 64, 8)       # [fragment][hi, lo][lane][j]
    ng = int(tile_ptr[-1]) // 2
    grp = np.zeros((max(ng, 1), 2), dtype=np.int32)
    chunks, units = [], 0
    for b in range(len(blk_row) - 1):
        rt = (int(blk_row[b + 1] - blk_row[b]) + 31) // 32
        base = int(tile_ptr[b])
        ngb = (int(tile_ptr[b + 1]) - base) // 2
        nq = ngb // rt if rt else 0
        for l in range(ngb):
            w, q = divmod(l, nq)
            f0 = base + (q * rt + w) * 2
            img = t[f0:f0 + 2]                                           # [k-step][hi, lo][lane][j]
            occ = (img[:, 0] | img[:, 1]) != 0                           # [k-step][lane][j]
            cells = int(occ.sum())
            g = base // 2 + l
            grp[g, 0] = units
            if cells > TILE_SPARSE_MAX:
                grp[g, 1] = -1
                chunks.append(np.ascontiguousarray(img).view(np.uint8).reshape(-1))
                units += 256
                continue
            grp[g, 1] = cells
            s, lane, j = np.nonzero(occ)                                 # image order
            ent = np.zeros((cells + (cells & 1), 2), dtype=np.uint32)
            ent[:cells, 0] = s * 2048 + lane * 16 + j * 2
            ent[:cells, 1] = img[s, 0, lane, j].astype(np.uint32) | (img[s, 1, lane, j].astype(np.uint32) << 16)
            chunks.append(ent.view(np.uint8).reshape(-1))
            units += ent.shape[0] // 2
    pool = np.concatenate(chunks) if chunks else np.zeros(0, dtype=np.uint8)
    return grp, pool


def tile_unpack(grp, pool, n_frag, blk_row, tile_ptr):
    """Inverse of tile_pack: the dense fragments (uint16 [n_frag * 1024]) rebuilt the way the kernel does — a cleared
    4 KiB image + the entries scattered (hi at the offset, lo 1024 bytes behind), or the copied image."""
    blk_row = np.asarray(blk_row, dtype=np.int64)
    tile_ptr = np.asarray(tile_ptr, dtype=np.int64)
    out = np.zeros((n_frag, 2048), dtype=np.uint8)
    pool = np.asarray(pool, dtype=np.uint8)
    for b in range(len(blk_row) - 1):
        rt = (int(blk_row[b + 1] - blk_row[b]) + 31) // 32
        base = int(tile_ptr[b])
        ngb = (int(tile_ptr[b + 1]) - base) // 2
        nq = ngb // rt if rt else 0
        for l in range(ngb):
            w, q = divmod(l, nq)
            f0 = base + (q * rt + w) * 2
            off, cells = int(np.uint32(grp[base // 2 + l, 0])) * 16, int(grp[base // 2 + l, 1])
            if cells < 0:
                out[f0:f0 + 2] = pool[off:off + 4096].reshape(2, 2048)
                continue
            img = np.zeros(4096, dtype=np.uint8).view(np.uint16)
            ent = pool[off:off + cells * 8].view(np.uint32).reshape(-1, 2)
            img[ent[:, 0] // 2] = (ent[:, 1] & 0xffff).astype(np.uint16)
            img[ent[:, 0] // 2 + 512] = (ent[:, 1] >> 16).astype(np.uint16)
            out[f0:f0 + 2] = img.view(np.uint8).reshape(2, 2048)
    return out.view(np.uint16).reshape(-1)


def spmm_tile(blk_row, sh_ptr, sh_cols, tile_ptr, tiles, rem_rowptr, rem_col, rem_val, x):
    """Y = A X evaluated THROUGH the tile plan (fragment cell -> slot -> staged source; remainder CSR), float64."""
    blk_row = np.asarray(blk_row, dtype=np.int64)
    x = np.asarray(x, dtype=np.float64)
    n = int(blk_row[-1])
    y = np.zeros((n, x.shape[1]), dtype=np.float64)
    t = np.asarray(tiles, dtype=np.uint16).reshape(-1, 2, 64, 8)
    a = bf16_bits_to_f32(t[:, 0]).astype(np.float64) + bf16_bits_to_f32(t[:, 1]).astype(np.float64)
    for b in range(blk_row.size - 1):
        rows = int(blk_row[b + 1] - blk_row[b])
        rt = (rows + 31) // 32
        s = int(sh_ptr[b + 1] - sh_ptr[b])
        cols = np.asarray(sh_cols[sh_ptr[b]:sh_ptr[b + 1]], dtype=np.int64)
        for q in range(s // 32):
            for ti in range(rt):
                for ks in range(2):
                    fr = a[int(tile_ptr[b]) + (q * rt + ti) * 2 + ks]          # [64 lanes][8]
                    for lane in range(64):
                        m = ti * 32 + (lane & 31)
                        if m >= rows:
                            continue
                        for j in range(8):
                            v = fr[lane, j]
                            if v != 0.0:
                                y[blk_row[b] + m] += v * x[cols[q * 32 + ks * 16 + 8 * (lane >> 5) + j]]
    for r in range(n):
        for e in range(int(rem_rowptr[r]), int(rem_rowptr[r + 1])):
            y[r] += float(rem_val[e]) * x[int(rem_col[e])]
    return y


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


_M64 = (1 << 64) - 1


def _mix64(z):
    """splitmix64 finaliser on Python ints (csrc/sampler.hip mix64)."""
    z = (z + 0x9E3779B97F4A7C15) & _M64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
    return z ^ (z >> 31)


def neighbor_sample(rowptr, colind, seeds, fanouts, seed, batch):
    """n_id, edge_src_local, edge_dst_local — sgf_neighbor_sample_* (csrc/sampler.hip), the draw included: hop h gives
    every node that entered in hop h - 1 min(in-degree, fanout) in-neighbours by Floyd's subset sampling (selection sampling
    above fan-out 32) with the
    counter-based hash of (seed, batch, hop, node); new nodes get local ids in order of first appearance."""
    rowptr = np.asarray(rowptr, dtype=np.int64)
    colind = np.asarray(colind, dtype=np.int64)
    n_id = [int(v) for v in seeds]
    local_of = {g: i for i, g in enumerate(n_id)}
    frontier, local0 = list(n_id), 0
    e_src, e_dst = [], []
    for hop, k in enumerate(fanouts):
        key = _mix64((seed & _M64) ^ _mix64((batch * 0x9E3779B97F4A7C15 + hop) & _M64))
        s_glob, dst = [], []
        for i, f in enumerate(frontier):
            base, deg = int(rowptr[f]), int(rowptr[f + 1] - rowptr[f])
            if k < 0 or deg <= k:
                picks = list(range(deg))
            else:
                nk = _mix64(key ^ ((f * 0xD6E8FEB86659FD93) & _M64))
                picks = []
                if k <= 32:          # Floyd's subset sampling, in draw order
                    for j, t in enumerate(range(deg - k, deg)):
                        r = _mix64((nk + j) & _M64) % (t + 1)
                        if r in picks:
                            r = t
                        picks.append(r)
                else:                # selection sampling (Knuth's algorithm S), in position order
                    for j in range(deg):
                        if len(picks) == k:
                            break
                        if _mix64((nk + j) & _M64) % (deg - j) < k - len(picks):
                            picks.append(j)
            s_glob += [int(colind[base + r]) for r in picks]
            dst += [local0 + i] * len(picks)
        new = []
        for g in s_glob:
            if g not in local_of:
                local_of[g] = len(n_id) + len(new)
                new.append(g)
        e_src += [local_of[g] for g in s_glob]
        e_dst += dst
        frontier, local0 = new, len(n_id)
        n_id += new
        if not frontier:
            break
    return (np.asarray(n_id, dtype=np.int64), np.asarray(e_src, dtype=np.int64), np.asarray(e_dst, dtype=np.int64))


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
