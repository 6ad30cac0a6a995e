"""The kernel table: one thin ctypes method per entry point of the libsgf C ABI (include/sgf.h), tensors in / tensors out,
on the current HIP stream.  Split out of ops.py in r05 (ops.py keeps the autograd-level operators built on this table).
PyTorch supplies memory and streams only; a CPU tensor or a missing library raises — there is no eager fallback.
"""
from __future__ import annotations

import ctypes
from typing import Optional

import torch

from . import _lib

_F32 = torch.float32
_BF16 = torch.bfloat16

# rows of the normalised adjacency longer than this are split into segments reduced by whole workgroups (sgf_spmm_split)
LONG_ROW = 1024
_SEGMENT = 1024   # = sgf_spmm_segment_len()


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


def _ld(t: Optional[torch.Tensor]) -> int:
    return 0 if t is None else t.stride(0)


def _rows(t: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    """A view/copy of a 2-D tensor whose rows are contiguous and 4-element aligned."""
    if t is None:
        return None
    if t.stride(-1) != 1 or t.stride(0) % 4 != 0 or t.data_ptr() % (4 * t.element_size()) != 0:
        t = t.contiguous()
        if t.shape[0] > 1 and t.stride(0) % 4 != 0:
            raise ValueError(f"feature dimension {t.shape[-1]} must be a multiple of 4")
    return t


_workspaces: "dict[tuple, torch.Tensor]" = {}


_capture_scope = None      # {(device index, name): tensor} while sgformer_amd.graphed captures a step, else None


def begin_capture_scope() -> dict:
    """Scratch requested while a step is being captured comes out of the graph's PRIVATE pool and its address is baked into
    the captured launches: it must live exactly as long as those graphs.  A process-wide cache keyed on the stream does the
    opposite — every capture runs on torch's one capture stream, so a later capture (another model, another batch size)
    would be handed scratch from an earlier graph's pool, and once that graph is destroyed the captured kernels write into
    unmapped memory (r06: `Memory access fault` in the eighth test of tests/test_gpu_graphed.py, never in isolation).  The
    caller keeps the returned dict with the captured step."""
    global _capture_scope
    _capture_scope = {}
    return _capture_scope


def end_capture_scope():
    global _capture_scope
    _capture_scope = None


def _workspace(device, name: str, nbytes: int) -> torch.Tensor:
    """Per-device, per-stream scratch reused across calls (users run on that stream, in order)."""
    if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
        if _capture_scope is None:           # somebody else's capture: fresh scratch from ITS pool, not cached here
            return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
        ws = _capture_scope.get((device.index, name))
        if ws is None or ws.numel() < nbytes:
            ws = _capture_scope[(device.index, name)] = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
        return ws
    key = (device.index, name, torch.cuda.current_stream(device).cuda_stream)
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
        _workspaces[key] = ws
    return ws


def _rows16(t: torch.Tensor) -> torch.Tensor:
    """Rows contiguous and 16-byte aligned (what the streaming row-GEMM loads as matrix-core fragments)."""
    if t.stride(-1) != 1 or (t.stride(0) * t.element_size()) % 16 != 0 or t.data_ptr() % 16 != 0:
        t = t.contiguous()
    return t



def _pair_gram() -> bool:
    import os
    return os.environ.get("SGF_GRAM_PAIR", "1") != "0"


def _one_pass_cat() -> bool:
    import os
    return os.environ.get("SGF_GCN_CAT", "1") != "0"



# ------------------------------------------------------------------------------------------------
# the kernel table: one method per C-ABI entry point, tensors in / tensors out
# ------------------------------------------------------------------------------------------------
class HipKernels:
    """libsgf.so on the current HIP stream.  Inputs must be GPU tensors."""

    name = "hip"

    @staticmethod
    def check(*tensors):
        for t in tensors:
            if t is not None and not t.is_cuda:
                raise RuntimeError(
                    "sgformer_amd runs on MI355X only: got a CPU tensor.  There is no CPU fallback; "
                    "move the model and its inputs to the GPU (the reference's CPU evaluation path, "
                    "large/eval.py:36-65, is outside this library's scope).")

    # ---- T1 ----
    @staticmethod
    def csr_build(ei: torch.Tensor, n: int):
        dev, nnz = ei.device, int(ei.shape[1])
        rowptr = torch.empty(n + 1, dtype=torch.int64, device=dev)
        colind = torch.empty(nnz, dtype=torch.int32, device=dev)
        val = torch.empty(nnz, dtype=_F32, device=dev)
        deg = torch.empty(n, dtype=torch.int32, device=dev)
        nbytes = _lib.load().sgf_csr_workspace_bytes(nnz, n)
        ws = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            _lib.call("sgf_csr_build", _ptr(ei), nnz, n, _ptr(rowptr), _ptr(colind), _ptr(val),
                      _ptr(deg), _ptr(ws), ws.numel(), _stream(dev))
        return rowptr, colind, val, deg

    @staticmethod
    def csr_transpose(ei: torch.Tensor, n: int, deg, rowptr, colind):
        dev, nnz = ei.device, int(ei.shape[1])
        t_rowptr = torch.empty(n + 1, dtype=torch.int64, device=dev)
        t_colind = torch.empty(nnz, dtype=torch.int32, device=dev)
        t_val = torch.empty(nnz, dtype=_F32, device=dev)
        flag = torch.zeros(1, dtype=torch.int32, device=dev)
        nbytes = _lib.load().sgf_csr_workspace_bytes(nnz, n)
        ws = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            _lib.call("sgf_csr_transpose", _ptr(ei), nnz, n, _ptr(deg), _ptr(rowptr), _ptr(colind),
                      _ptr(t_rowptr), _ptr(t_colind), _ptr(t_val), _ptr(flag), _ptr(ws), ws.numel(),
                      _stream(dev))
        return t_rowptr, t_colind, t_val, bool(int(flag.item()))  # one host sync, once per graph

    # ---- N1: induced subgraph ----
    @staticmethod
    def subgraph(ei: torch.Tensor, n: int, subset: torch.Tensor, relabel_nodes: bool, want_eid: bool):
        """ei int64 [2, nnz] and subset int64 [m] on the GPU -> (edge_index_sub [2, k], eid [k] | None)."""
        dev, nnz, m = ei.device, int(ei.shape[1]), int(subset.numel())
        lib = _lib.load()
        relabel = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
        total = torch.zeros(1, dtype=torch.int64, device=dev)
        ws = _workspace(dev, "subgraph", lib.sgf_subgraph_workspace_bytes(nnz, n))
        with torch.cuda.device(dev):
            _lib.call("sgf_subgraph_plan", _ptr(ei), nnz, n, _ptr(subset), m, _ptr(relabel), _ptr(total),
                      _ptr(ws), ws.numel(), _stream(dev))
            k = int(total.item())              # the one host sync: the output size
            out = torch.empty((2, k), dtype=torch.int64, device=dev)
            eid = torch.empty(k, dtype=torch.int64, device=dev) if want_eid else None
            _lib.call("sgf_subgraph_emit", _ptr(ei), nnz, n, _ptr(relabel), int(relabel_nodes), k, _ptr(out),
                      _ptr(eid), _ptr(ws), ws.numel(), _stream(dev))
        return out, eid

    @staticmethod
    def subgraph_csr(rowptr, colind, n: int, subset: torch.Tensor, local_of: torch.Tensor, want_edges: bool):
        """Induced subgraph + its normalised CSR from the parent CSR (sgf_subgraph_csr_*): (rowptr_b, colind_b, val_b, deg_b,
        edge_index_b | None, longest row), or None when `subset` repeats a node / leaves the graph (the caller takes
        sgf_subgraph_*)."""
        dev, m = rowptr.device, int(subset.numel())
        lib = _lib.load()
        rowptr_b = torch.empty(m + 1, dtype=torch.int64, device=dev)
        deg_b = torch.empty(m + 1, dtype=torch.int32, device=dev)
        total = torch.empty(3, dtype=torch.int64, device=dev)
        ws = _workspace(dev, "subgraph_csr_plan", lib.sgf_subgraph_csr_plan_workspace_bytes(m))
        with torch.cuda.device(dev):
            _lib.call("sgf_subgraph_csr_plan", _ptr(rowptr), _ptr(colind), n, _ptr(subset), m, _ptr(local_of), _ptr(rowptr_b),
                      _ptr(deg_b), _ptr(total), _ptr(ws), ws.numel(), _stream(dev))
            # the plan MARKED the parent's shared `local_of` table; only the emit call clears the marks again.  Whatever fails
            # in between (the host read, an allocation) must not leave them behind: later batches would pick up stale local
            # ids for nodes outside their subset.
            try:
                t, bad, longest = total.tolist()            # the one host read of the batch
                t = 0 if bad else int(t)
                colind_b = torch.empty(t, dtype=torch.int32, device=dev)
                val_b = torch.empty(t, dtype=_F32, device=dev)
                ei_b = torch.empty((2, t), dtype=torch.int64, device=dev) if want_edges else None
                ws2 = _workspace(dev, "subgraph_csr_emit", lib.sgf_subgraph_csr_emit_workspace_bytes(m, t))
            except BaseException:
                local_of.fill_(-1)
                raise
            _lib.call("sgf_subgraph_csr_emit", _ptr(rowptr), _ptr(colind), n, _ptr(subset), m, _ptr(local_of), _ptr(rowptr_b),
                      _ptr(deg_b), t, _ptr(colind_b), _ptr(val_b), _ptr(ei_b), _ptr(ws2), ws2.numel(), _stream(dev))
        if bad:
            return None
        return rowptr_b, colind_b, val_b, deg_b[:m], ei_b, int(longest)

    # ---- N2: trainer prologue (to_undirected / remove_self_loops / add_self_loops) ----
    @staticmethod
    def graph_prologue(ei: torch.Tensor, n: int, undirected: bool, remove_loops: bool, add_loops: bool):
        dev, m = ei.device, int(ei.shape[1])
        lib = _lib.load()
        total = torch.zeros(1, dtype=torch.int64, device=dev)
        ws = torch.empty(max(lib.sgf_graph_prologue_workspace_bytes(m, n), 256), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            _lib.call("sgf_graph_prologue_plan", _ptr(ei), m, n, int(undirected), int(remove_loops), int(add_loops),
                      _ptr(total), _ptr(ws), ws.numel(), _stream(dev))
            k = int(total.item())              # the one host sync: the output size
            out = torch.empty((2, k), dtype=torch.int64, device=dev)
            _lib.call("sgf_graph_prologue_emit", m, n, int(undirected), int(add_loops), k, _ptr(out), _ptr(ws),
                      ws.numel(), _stream(dev))
        return out

    # ---- T2 ----
    @staticmethod
    def spmm(rowptr, colind, val, x: torch.Tensor, n_rows: int, out: Optional[torch.Tensor] = None,
             long_segments: int = 0, stream_hint: bool = False) -> torch.Tensor:
        """`out`: optional [n_rows, d] destination, possibly a column slice of a wider buffer.
        `long_segments` > 0 (see long_row_segments): rows longer than LONG_ROW are split across
        workgroups (sgf_spmm_split).  `stream_hint`: the CSR's gathers mostly hit in L2 (re-ordered graph):
        sgf_spmm_stream."""
        x = _rows(x)
        d = x.shape[1]
        y = torch.empty((n_rows, d), dtype=x.dtype, device=x.device) if out is None else out
        if out is not None and (out.shape != (n_rows, d) or out.dtype != x.dtype or out.stride(1) != 1
                                or out.stride(0) % 4 != 0 or out.data_ptr() % (4 * out.element_size()) != 0):
            raise ValueError("spmm: `out` must be [n_rows, d], same dtype, row-contiguous and 4-element aligned")
        if n_rows == 0 or d == 0:
            return y
        with torch.cuda.device(x.device):
            if long_segments > 0 or stream_hint:
                ws = (_workspace(x.device, "spmm_long", _lib.load().sgf_spmm_split_workspace_bytes(long_segments, d))
                      if long_segments > 0 else None)
                _lib.call("sgf_spmm_stream" if stream_hint else "sgf_spmm_split", _ptr(rowptr), _ptr(colind), _ptr(val), _ptr(x), x.stride(0), x.shape[0],
                          _ptr(y), y.stride(0), n_rows, d, _code(x), LONG_ROW, long_segments, _ptr(ws),
                          0 if ws is None else ws.numel(), _stream(x.device))
            else:
                _lib.call("sgf_spmm", _ptr(rowptr), _ptr(colind), _ptr(val), _ptr(x), x.stride(0), x.shape[0],
                          _ptr(y), y.stride(0), n_rows, d, _code(x), _stream(x.device))
        return y

    # ---- T2 with on-chip reuse: node order, row-block plan, LDS-staged SpMM ----
    @staticmethod
    def reorder(ei: torch.Tensor, n: int, iters1: int, iters2: int):
        """perm (new position -> old id), inv (old id -> new position), community: int32 [n] each."""
        dev, nnz = ei.device, int(ei.shape[1])
        perm = torch.empty(n, dtype=torch.int32, device=dev)
        inv = torch.empty(n, dtype=torch.int32, device=dev)
        comm = torch.empty(n, dtype=torch.int32, device=dev)
        nbytes = _lib.load().sgf_reorder_workspace_bytes(nnz, n)
        ws = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            _lib.call("sgf_reorder", _ptr(ei), nnz, n, int(iters1), int(iters2), _ptr(perm), _ptr(inv),
                      _ptr(comm), _ptr(ws), ws.numel(), _stream(dev))
        return perm, inv, comm

    @staticmethod
    def spmm_plan(rowptr, colind, val, n: int, rows_per_block: int, lds_rows: int, long_len: int):
        dev, nnz = rowptr.device, int(colind.numel())
        nb = (n + rows_per_block - 1) // rows_per_block
        ecode = torch.empty(nnz, dtype=torch.int32, device=dev)
        ev = torch.empty(nnz, dtype=_F32, device=dev)
        nlds = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
        sh_ptr = torch.empty(nb + 1, dtype=torch.int32, device=dev)
        sh_cols = torch.empty(max(nb * lds_rows, 1), dtype=torch.int32, device=dev)
        stats = torch.zeros(4, dtype=torch.int64, device=dev)
        nbytes = _lib.load().sgf_spmm_plan_workspace_bytes(nnz, n, rows_per_block)
        ws = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            _lib.call("sgf_spmm_plan", _ptr(rowptr), _ptr(colind), _ptr(val), n, nnz, int(rows_per_block),
                      int(lds_rows), int(long_len), _ptr(ecode), _ptr(ev), _ptr(nlds), _ptr(sh_ptr),
                      _ptr(sh_cols), _ptr(stats), _ptr(ws), ws.numel(), _stream(dev))
        st = [int(v) for v in stats.tolist()]                      # one host sync, once per plan
        return ecode, ev, nlds, sh_ptr, sh_cols[: max(st[1], 1)].clone(), st

    @staticmethod
    def spmm_blocked(rowptr, plan, x: torch.Tensor, n_rows: int, out: Optional[torch.Tensor] = None,
                     long_segments: int = 0) -> torch.Tensor:
        x = _rows(x)
        d = x.shape[1]
        y = torch.empty((n_rows, d), dtype=x.dtype, device=x.device) if out is None else out
        if n_rows == 0 or d == 0:
            return y
        with torch.cuda.device(x.device):
            ws = None
            if long_segments > 0:
                ws = _workspace(x.device, "spmm_long", _lib.load().sgf_spmm_split_workspace_bytes(long_segments, d))
            _lib.call("sgf_spmm_blocked", _ptr(rowptr), _ptr(plan.ecode), _ptr(plan.eval), _ptr(plan.nlds),
                      _ptr(plan.sh_ptr), _ptr(plan.sh_cols), _ptr(x), x.stride(0), _ptr(y), y.stride(0), n_rows, d,
                      _code(x), plan.rows_per_block, plan.lds_rows, LONG_ROW, long_segments, _ptr(ws),
                      0 if ws is None else ws.numel(), _stream(x.device))
        return y

    # ---- T2 on a re-ordered graph: dense matrix-core tiles + gather remainder (csrc/spmm_tile.hip) ----
    @staticmethod
    def tile_supported(d: int, dtype) -> bool:
        return bool(_lib.load().sgf_spmm_tile_supported(int(d), _lib.SGF_BF16 if dtype == _BF16 else _lib.SGF_F32))

    @staticmethod
    def tile_blocks(comm_sorted: Optional[torch.Tensor], n: int, max_rows: int, device) -> torch.Tensor:
        """blk_row int32 [nb + 1]: row blocks that follow the communities (comm_sorted[p] = community of new row p)."""
        cap = 4 * n // int(max_rows) + 4
        blk = torch.empty(cap + 1, dtype=torch.int32, device=device)
        nb = ctypes.c_int64(0)
        with torch.cuda.device(device):
            _lib.call("sgf_spmm_tile_blocks", _ptr(comm_sorted), n, int(max_rows), _ptr(blk), cap, ctypes.byref(nb),
                      _stream(device))
        return blk[: nb.value + 1].clone()

    @staticmethod
    def tile_plan(rowptr, colind, val, n: int, blk_row: torch.Tensor, cap: int, min_count: int, long_len: int):
        """(sh_ptr, sh_cols, tile_ptr, tiles, rem_rowptr, rem_col, rem_val, stats) — see include/sgf.h."""
        dev, nnz, nb = rowptr.device, int(colind.numel()), int(blk_row.numel()) - 1
        ecode = torch.empty(max(nnz, 1), dtype=torch.int32, device=dev)
        ev = torch.empty(max(nnz, 1), dtype=_F32, device=dev)
        nlds = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
        sh_ptr = torch.empty(nb + 1, dtype=torch.int32, device=dev)
        sh_cols = torch.empty(max(nb * cap, 1), dtype=torch.int32, device=dev)
        tile_ptr = torch.empty(nb + 1, dtype=torch.int64, device=dev)
        rem_rowptr = torch.empty(n + 1, dtype=torch.int64, device=dev)
        stats = torch.zeros(8, dtype=torch.int64, device=dev)
        nbytes = _lib.load().sgf_spmm_tile_plan_workspace_bytes(nnz, n, nb)
        ws = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            _lib.call("sgf_spmm_tile_plan", _ptr(rowptr), _ptr(colind), _ptr(val), n, nnz, _ptr(blk_row), nb, int(cap),
                      int(min_count), int(long_len), _ptr(ecode), _ptr(ev), _ptr(nlds), _ptr(sh_ptr), _ptr(sh_cols),
                      _ptr(tile_ptr), _ptr(rem_rowptr), _ptr(stats), _ptr(ws), ws.numel(), _stream(dev))
            st = [int(v) for v in stats.tolist()]                      # one host sync, once per plan
            del ws
            if st[7]:
                raise ValueError("tile_plan: a row block is empty or longer than 256 rows")
            n_frag, n_rem = st[4], st[5]
            tiles = torch.empty(max(n_frag, 1) * 512, dtype=torch.int32, device=dev)     # 2 KiB per fragment
            rem_col = torch.empty(max(n_rem, 1), dtype=torch.int32, device=dev)
            rem_val = torch.empty(max(n_rem, 1), dtype=_F32, device=dev)
            _lib.call("sgf_spmm_tile_fill", _ptr(rowptr), _ptr(ecode), _ptr(ev), _ptr(nlds), n, nnz, _ptr(blk_row), nb,
                      _ptr(tile_ptr), n_frag, _ptr(rem_rowptr), _ptr(tiles), _ptr(rem_col), _ptr(rem_val), _stream(dev))
        return sh_ptr, sh_cols[: max(st[1], 1)].clone(), tile_ptr, tiles, rem_rowptr, rem_col, rem_val, st

    @staticmethod
    def tile_pack(blk_row: torch.Tensor, tile_ptr: torch.Tensor, tiles: torch.Tensor, n_frag: int):
        """(grp int32 [n_frag / 2, 2], pool uint8): the fragments as the kernel streams them — sparse groups as 8-byte
        entries, dense ones as they are (csrc/spmm_pack.hip, include/sgf.h)."""
        dev, nb = tiles.device, int(blk_row.numel()) - 1
        ng = n_frag // 2
        grp = torch.zeros((max(ng, 1), 2), dtype=torch.int32, device=dev)
        units = torch.zeros(1, dtype=torch.int64, device=dev)
        ws = torch.empty(max(_lib.load().sgf_spmm_tile_pack_workspace_bytes(n_frag), 256), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            _lib.call("sgf_spmm_tile_pack_layout", _ptr(blk_row), nb, _ptr(tile_ptr), _ptr(tiles), n_frag, _ptr(grp),
                      _ptr(units), _ptr(ws), ws.numel(), _stream(dev))
            nu = int(units.item())                                      # one host sync, once per plan
            pool = torch.zeros(nu * 16 + 4096, dtype=torch.uint8, device=dev)   # the kernel fetches whole KiB
            _lib.call("sgf_spmm_tile_pack", _ptr(blk_row), nb, _ptr(tile_ptr), _ptr(tiles), n_frag, _ptr(grp), _ptr(pool),
                      nu, _stream(dev))
        return grp, pool, nu

    @staticmethod
    def spmm_tile(plan, x: torch.Tensor, n_rows: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        x = _rows16(x)
        d = x.shape[1]
        y = torch.empty((n_rows, d), dtype=x.dtype, device=x.device) if out is None else out
        if n_rows == 0 or d == 0:
            return y
        segs = plan.long_segments
        with torch.cuda.device(x.device):
            ws = None
            if segs > 0:
                ws = _workspace(x.device, "spmm_long", _lib.load().sgf_spmm_split_workspace_bytes(segs, d))
            _lib.call("sgf_spmm_tile", _ptr(plan.blk_row), plan.nb, plan.block_rows, _ptr(plan.sh_ptr), _ptr(plan.sh_cols),
                      _ptr(plan.tile_ptr), _ptr(plan.grp), _ptr(plan.pool), _ptr(plan.rem_rowptr), _ptr(plan.rem_col),
                      _ptr(plan.rem_val), _ptr(x), x.stride(0), x.shape[0], _ptr(y), y.stride(0), n_rows, d, _code(x),
                      LONG_ROW, segs, _ptr(ws), 0 if ws is None else ws.numel(), _stream(x.device))
        return y

    @staticmethod
    def lds_rows_max(dtype) -> int:
        return int(_lib.load().sgf_spmm_lds_rows_len(_lib.SGF_BF16 if dtype == _BF16 else _lib.SGF_F32))

    @staticmethod
    def gather_rows(src: torch.Tensor, idx: torch.Tensor, out_dtype=None) -> torch.Tensor:
        """out[i] = src[idx[i]] (idx int32 / int64 on the GPU), optionally cast fp32 <-> bf16 on the way."""
        if src.stride(-1) != 1:
            src = src.contiguous()
        n_out, d = int(idx.numel()), src.shape[1]
        out = torch.empty((n_out, d), dtype=out_dtype or src.dtype, device=src.device)
        with torch.cuda.device(src.device):
            _lib.call("sgf_gather_rows", _ptr(src), src.stride(0), _code(src), src.shape[0], _ptr(idx),
                      int(idx.dtype == torch.int64), n_out, d, _ptr(out), out.stride(0), _code(out),
                      _stream(src.device))
        return out

    @staticmethod
    def pad_rows(src: torch.Tensor, idx: Optional[torch.Tensor], d_pad: int, out_dtype=None) -> torch.Tensor:
        """out[i, :d] = src[idx[i] if idx is given else i, :d], out[i, d:d_pad] = 0, optionally cast fp32 <-> bf16 — the
        aligned copy of features whose width is not a multiple of 4 (sgf_pad_rows)."""
        if src.stride(-1) != 1:
            src = src.contiguous()
        n_out = int(src.shape[0] if idx is None else idx.numel())
        d = src.shape[1]
        out = torch.empty((n_out, d_pad), dtype=out_dtype or src.dtype, device=src.device)
        with torch.cuda.device(src.device):
            _lib.call("sgf_pad_rows", _ptr(src), src.stride(0), _code(src), src.shape[0], _ptr(idx),
                      int(idx is not None and idx.dtype == torch.int64), n_out, d, d_pad, _ptr(out), out.stride(0),
                      _code(out), _stream(src.device))
        return out

    # ---- T4: the general Linear (any shape, any alignment; csrc/gemm.hip) ----
    @staticmethod
    def gemm(a: torch.Tensor, b: torch.Tensor, bias=None, out=None, out_dtype=None, alpha: float = 1.0, alpha_dev=None,
             beta: float = 0.0, addend=None) -> torch.Tensor:
        """out[m, n] = alpha * a[m, k] @ b[k, n] + bias + beta * addend — a, b any 2-D VIEWS (strides are passed on, so
        x @ w.t() costs no copy), fp32 or bf16 storage each; out: given, or new in out_dtype (default: a's dtype)."""
        m, k = a.shape
        k2, n = b.shape
        if k != k2:
            raise RuntimeError(f"gemm: inner dimensions differ: {tuple(a.shape)} x {tuple(b.shape)}")
        dev = a.device
        if out is None:
            out = torch.empty((m, n), dtype=out_dtype or a.dtype, device=dev)
        elif out.stride(-1) != 1 and n > 1:
            raise RuntimeError("gemm: out must have contiguous rows")
        if addend is not None and (addend.stride(-1) != 1 and n > 1):
            addend = addend.contiguous()
        with torch.cuda.device(dev):
            _lib.call("sgf_gemm", _ptr(a), a.stride(0), a.stride(1), _code(a), _ptr(b), b.stride(0), b.stride(1), _code(b),
                      m, n, k, float(alpha), _ptr(alpha_dev), _ptr(bias), float(beta), _ptr(addend),
                      0 if addend is None else max(addend.stride(0), n), 0 if addend is None else _code(addend),
                      _ptr(out), max(out.stride(0), n), _code(out), _stream(dev))
        return out

    # ---- T3: the d x d algebra of the attention as one call each way (csrc/attn_small.hip) ----
    @staticmethod
    def attn_h_small_fwd(G, s, n_rows: float, n_total: float, wq, bq, wk, bk, wv, bv):
        """(M [D, d], m [d], w [D], beta [1], saved) from the Gram matrix G = h^T h [D, D], s = sum_n h_n and the three
        projections' fp32 weights [d, D] / biases [d] (wv None: V = h)."""
        d, D = wq.shape
        dev = G.device
        lib = _lib.load()
        ws = [t if (t is None or (t.dtype == _F32 and t.is_contiguous())) else t.float().contiguous()
              for t in (wq, bq, wk, bk, wv, bv)]
        G = G if (G.dtype == _F32 and G.stride(-1) == 1) else G.float().contiguous()
        s = s if (s.dtype == _F32 and s.is_contiguous()) else s.float().contiguous()
        M = torch.empty((D, d), dtype=_F32, device=dev)
        mwb = torch.empty(d + D + 1, dtype=_F32, device=dev)
        saved = torch.empty(lib.sgf_attn_h_small_saved_bytes(D, d), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            _lib.call("sgf_attn_h_small_fwd", _ptr(G), G.stride(0), _ptr(s), float(n_rows), float(n_total), _ptr(ws[0]),
                      _ptr(ws[1]), _ptr(ws[2]), _ptr(ws[3]), _ptr(ws[4]), _ptr(ws[5]), D, D, d, _ptr(M), d,
                      _ptr(mwb[:d]), _ptr(mwb[d:d + D]), _ptr(mwb[d + D:]), _ptr(saved), saved.numel(), _stream(dev))
        return M, mwb[:d], mwb[d:d + D], mwb[d + D:], saved

    @staticmethod
    def attn_h_small_bwd(dM, dw, dm, dbeta, n_total: float, saved, d_in: int, d_out: int, want_v: bool = True):
        """(D = dG + dG^T [D, D], ds [D], gwq, gbq, gwk, gbk, gwv | None, gbv | None) from the reduced gradients of
        M, w, m, beta (the blocks of hstats) and what attn_h_small_fwd saved."""
        D, d = d_in, d_out
        dev = dM.device
        lib = _lib.load()
        ws = _workspace(dev, "attn_small", lib.sgf_attn_h_small_workspace_bytes(D, d))
        Dm = torch.empty((D, D), dtype=_F32, device=dev)
        ds = torch.empty(D, dtype=_F32, device=dev)
        gw = torch.empty((3 if want_v else 2, d, D), dtype=_F32, device=dev)
        gb = torch.empty((3 if want_v else 2, d), dtype=_F32, device=dev)
        with torch.cuda.device(dev):
            _lib.call("sgf_attn_h_small_bwd", _ptr(dM), dM.stride(0), _ptr(dw), _ptr(dm), _ptr(dbeta), float(n_total), D, d,
                      _ptr(saved), saved.numel(), _ptr(Dm), D, _ptr(ds), _ptr(gw[0]), _ptr(gb[0]), _ptr(gw[1]), _ptr(gb[1]),
                      _ptr(gw[2]) if want_v else None, _ptr(gb[2]) if want_v else None, D, _ptr(ws), ws.numel(),
                      _stream(dev))
        return Dm, ds, gw[0], gb[0], gw[1], gb[1], (gw[2] if want_v else None), (gb[2] if want_v else None)

    # ---- T3 ----  q, k: [n, H*d] views (ld = stride(0)); v: [n, Hv*d]
    @staticmethod
    def attn_fwd_reduce(q, k, v, heads: int, v_heads: int, d: int) -> torch.Tensor:
        n, dev = q.shape[0], q.device
        lib = _lib.load()
        stats = torch.empty(lib.sgf_attn_stats_len(heads, d), dtype=_F32, device=dev)
        ws = _workspace(dev, "attn", lib.sgf_attn_workspace_bytes(n, heads, d))
        with torch.cuda.device(dev):
            _lib.call("sgf_attn_fwd_reduce", _ptr(q), _ld(q), _ptr(k), _ld(k), _ptr(v), _ld(v), n,
                      heads, v_heads, d, _code(q), _ptr(stats), _ptr(ws), ws.numel(), _stream(dev))
        return stats

    @staticmethod
    def attn_fwd_apply(q, v, stats, n_total: float, heads: int, v_heads: int, d: int):
        n, dev = q.shape[0], q.device
        out = torch.empty((n, d), dtype=q.dtype, device=dev)
        den = torch.empty((n, heads), dtype=_F32, device=dev)
        o_heads = torch.empty((n, heads * d), dtype=q.dtype, device=dev) if heads > 1 else None
        with torch.cuda.device(dev):
            _lib.call("sgf_attn_fwd_apply", _ptr(q), _ld(q), _ptr(v), _ld(v), n, float(n_total),
                      heads, v_heads, d, _code(q), _ptr(stats), _ptr(out), out.stride(0), _ptr(den),
                      _ptr(o_heads), _stream(dev))
        return out, den, o_heads

    @staticmethod
    def attn_bwd_reduce(q, g, o, den, heads: int, d: int, per_head: bool = False) -> torch.Tensor:
        """per_head: g is [n, H * d], the gradients of the per-head outputs (no 1/H), instead of the head mean's [n, d]"""
        n, dev = q.shape[0], q.device
        lib = _lib.load()
        bstats = torch.empty(lib.sgf_attn_bstats_len(heads, d), dtype=_F32, device=dev)
        ws = _workspace(dev, "attn", lib.sgf_attn_workspace_bytes(n, heads, d))
        with torch.cuda.device(dev):
            _lib.call("sgf_attn_bwd_reduce_heads" if per_head else "sgf_attn_bwd_reduce", _ptr(q), _ld(q), _ptr(g), _ld(g), _ptr(o), _ld(o),
                      _ptr(den), n, heads, d, _code(q), _ptr(bstats), _ptr(ws), ws.numel(),
                      _stream(dev))
        return bstats

    @staticmethod
    def attn_bwd_apply(q, k, v, g, o, den, stats, bstats, n_total: float, heads: int, v_heads: int,
                       d: int, dq, dk, dv, per_head: bool = False):
        """Writes dq, dk, dv (views with row stride = stride(0)) in place."""
        n, dev = q.shape[0], q.device
        with torch.cuda.device(dev):
            _lib.call("sgf_attn_bwd_apply_heads" if per_head else "sgf_attn_bwd_apply", _ptr(q), _ld(q), _ptr(k), _ld(k), _ptr(v), _ld(v), _ptr(g),
                      _ld(g), _ptr(o), _ld(o), _ptr(den), n, float(n_total), heads, v_heads, d,
                      _code(q), _ptr(stats), _ptr(bstats), _ptr(dq), _ld(dq), _ptr(dk), _ld(dk),
                      _ptr(dv), _ld(dv), _stream(dev))

    # ---- T3+T4 fused: attention from the un-projected input (H = 1) ----
    @staticmethod
    def attn_h_fwd(h, M, m, w, beta):
        n, d = h.shape
        dev = h.device
        out = torch.empty((n, d), dtype=h.dtype, device=dev)
        den = torch.empty((n, 1), dtype=_F32, device=dev)
        with torch.cuda.device(dev):
            _lib.call("sgf_attn_h_fwd", _ptr(h), _ld(h), n, d, _code(h), _ptr(M), _ptr(m), _ptr(w),
                      _ptr(beta), _ptr(out), out.stride(0), _ptr(den), _stream(dev))
        return out, den

    @staticmethod
    def attn_h_bwd_reduce(h, g, o, den):
        n, d = h.shape
        dev = h.device
        lib = _lib.load()
        hstats = torch.empty(lib.sgf_attn_h_bstats_len(d), dtype=_F32, device=dev)
        ws = _workspace(dev, "attn", lib.sgf_attn_workspace_bytes(n, 1, d))
        with torch.cuda.device(dev):
            _lib.call("sgf_attn_h_bwd_reduce", _ptr(h), _ld(h), _ptr(g), _ld(g), _ptr(o), _ld(o),
                      _ptr(den), n, d, _code(h), _ptr(hstats), _ptr(ws), ws.numel(), _stream(dev))
        return hstats

    @staticmethod
    def attn_h_bwd_apply(h, g, o, den, M, w, D, ds):
        n, d = h.shape
        dev = h.device
        dh = torch.empty((n, d), dtype=h.dtype, device=dev)
        nb = _lib.load().sgf_attn_h_bwd_apply_workspace_bytes(n, d, _code(h))
        ws = _workspace(dev, "attn_h_part", nb) if nb else None
        with torch.cuda.device(dev):
            _lib.call("sgf_attn_h_bwd_apply", _ptr(h), _ld(h), _ptr(g), _ld(g), _ptr(o), _ld(o),
                      _ptr(den), n, d, _code(h), _ptr(M), _ptr(w), _ptr(D), _ptr(ds), _ptr(dh),
                      dh.stride(0), _ptr(ws), 0 if ws is None else ws.numel(), _stream(dev))
        return dh

    @staticmethod
    def attn_h_bwd_split_supported(h, g, o) -> bool:
        d = h.shape[1]
        return (h.dtype == _BF16 and bool(_lib.load().sgf_attn_h_bwd_split_supported(d, _lib.SGF_BF16))
                and all(t.stride(-1) == 1 and (t.stride(0) * 2) % 16 == 0 and t.data_ptr() % 16 == 0 for t in (h, g, o)))

    @staticmethod
    def attn_h_bwd_pre(g, o, den, M, w):
        """First apply pass of the backward (dnum M^T + dden w -> scratch) + the per-row scalars (1/den, dden)."""
        n, d = g.shape
        dev = g.device
        ws = _workspace(dev, "attn_h_part", _lib.load().sgf_attn_h_bwd_apply_workspace_bytes(n, d, _code(g)))
        rowscal = torch.empty((n, 2), dtype=_F32, device=dev)
        with torch.cuda.device(dev):
            _lib.call("sgf_attn_h_bwd_pre", _ptr(g), _ld(g), _ptr(o), _ld(o), _ptr(den), n, d, _code(g), _ptr(M),
                      _ptr(w), _ptr(ws), ws.numel(), _ptr(rowscal), _stream(dev))
        return rowscal

    @staticmethod
    def attn_h_bwd_reduce_scaled(h, g, rowscal):
        n, d = h.shape
        dev = h.device
        lib = _lib.load()
        hstats = torch.empty(lib.sgf_attn_h_bstats_len(d), dtype=_F32, device=dev)
        ws = _workspace(dev, "attn", lib.sgf_attn_workspace_bytes(n, 1, d))
        with torch.cuda.device(dev):
            _lib.call("sgf_attn_h_bwd_reduce_scaled", _ptr(h), _ld(h), _ptr(g), _ld(g), _ptr(rowscal), n, d, _code(h),
                      _ptr(hstats), _ptr(ws), ws.numel(), _stream(dev))
        return hstats

    @staticmethod
    def attn_h_bwd_post(h, D, ds, addend=None):
        """dh = h D + ds + the scratch attn_h_bwd_pre left on this stream [+ addend, a second gradient of h]."""
        n, d = h.shape
        dev = h.device
        ws = _workspace(dev, "attn_h_part", _lib.load().sgf_attn_h_bwd_apply_workspace_bytes(n, d, _code(h)))
        dh = torch.empty((n, d), dtype=h.dtype, device=dev)
        with torch.cuda.device(dev):
            _lib.call("sgf_attn_h_bwd_post", _ptr(h), _ld(h), n, d, _code(h), _ptr(D), _ptr(ds), _ptr(ws), ws.numel(),
                      _ptr(addend), 0 if addend is None else _ld(addend), _ptr(dh), dh.stride(0), _stream(dev))
        return dh

    # ---- T4: dW = a^T b, db = colsum(a) ----
    @staticmethod
    def gram(a, b, out=None, want_colsum=True):
        """out[m, k] (fp32, may be a column-sliced view) = a[n, m]^T b[n, k]; colsum(a) fp32 [m]."""
        n, m = a.shape
        k = b.shape[1]
        dev = a.device
        if out is None:
            out = torch.empty((m, k), dtype=_F32, device=dev)
        cs = torch.empty(m, dtype=_F32, device=dev) if want_colsum else None
        ws = _workspace(dev, "attn", _lib.load().sgf_gram_workspace_bytes(n, m, k))
        with torch.cuda.device(dev):
            _lib.call("sgf_gram", _ptr(a), _ld(a), m, _ptr(b), _ld(b), k, n, _code(a), _ptr(out),
                      out.stride(0), _ptr(cs), _ptr(ws), ws.numel(), _stream(dev))
        return out, cs

    @staticmethod
    def gram2(a, b1, b2, out1, out2, want_colsum=True):
        """out1 = a^T b1, out2 = a^T b2 (fp32, may be column-sliced views of one matrix) from one paired launch; colsum(a)."""
        n, m = a.shape
        k = b1.shape[1]
        dev = a.device
        cs = torch.empty(m, dtype=_F32, device=dev) if want_colsum else None
        ws = _workspace(dev, "attn", _lib.load().sgf_gram_workspace_bytes(n, m, k))
        with torch.cuda.device(dev):
            _lib.call("sgf_gram2", _ptr(a), _ld(a), m, _ptr(b1), _ld(b1), _ptr(b2), _ld(b2), k, n, _code(a), _ptr(out1),
                      out1.stride(0), _ptr(out2), out2.stride(0), _ptr(cs), _ptr(ws), ws.numel(), _stream(dev))
        return cs

    # ---- T5 ----
    @staticmethod
    def ln_fwd(x, res, a: float, b: float, gamma, beta, relu: bool, eps: float):
        n, d = x.shape
        dev = x.device
        y = torch.empty((n, d), dtype=x.dtype, device=dev)
        mean = torch.empty(n, dtype=_F32, device=dev) if gamma is not None else None
        rstd = torch.empty(n, dtype=_F32, device=dev) if gamma is not None else None
        with torch.cuda.device(dev):
            _lib.call("sgf_ln_fwd", _ptr(x), _ld(x), _ptr(res), _ld(res), float(a), float(b),
                      _ptr(gamma), _ptr(beta), int(relu), float(eps), n, d, _code(x), _ptr(y),
                      y.stride(0), _ptr(mean), _ptr(rstd), _stream(dev))
        return y, mean, rstd

    @staticmethod
    def ln_bwd(gy, y, x, res, a: float, b: float, gamma, relu: bool, mean, rstd):
        """(dx, dres, dgamma, dbeta).  a == b (the large variant's (x + res) / 2): dx and dres are the same values —
        ONE tensor is written and returned for both."""
        n, d = x.shape
        dev = x.device
        dx = torch.empty_like(x)
        shared = res is not None and float(a) == float(b)
        dres = torch.empty_like(res) if (res is not None and not shared) else None
        dgamma = torch.empty(d, dtype=_F32, device=dev) if gamma is not None else None
        dbeta = torch.empty(d, dtype=_F32, device=dev) if gamma is not None else None
        ws = _workspace(dev, "ln", _lib.load().sgf_ln_bwd_workspace_bytes(n, d))
        with torch.cuda.device(dev):
            _lib.call("sgf_ln_bwd", _ptr(gy), _ld(gy), _ptr(y), _ld(y), _ptr(x), _ld(x), _ptr(res),
                      _ld(res), float(a), float(b), _ptr(gamma), int(relu), _ptr(mean), _ptr(rstd), n,
                      d, _code(x), _ptr(dx), _ld(dx), _ptr(dres), _ld(dres), _ptr(dgamma), _ptr(dbeta),
                      _ptr(ws), ws.numel(), _stream(dev))
        return dx, (dx if shared else dres), dgamma, dbeta

    # ---- T6 ----
    @staticmethod
    def colstats(x, shift) -> torch.Tensor:
        n, d = x.shape
        dev = x.device
        stats = torch.empty(2 * d, dtype=_F32, device=dev)
        ws = _workspace(dev, "col", _lib.load().sgf_colstats_workspace_bytes(n, d))
        with torch.cuda.device(dev):
            _lib.call("sgf_colstats", _ptr(x), _ld(x), _ptr(shift), n, d, _code(x), _ptr(stats),
                      _ptr(ws), ws.numel(), _stream(dev))
        return stats

    @staticmethod
    def bn_finalize(sums, shift, n_total: float, eps: float, momentum: float, running_mean, running_var):
        """(mean, rstd) fp32 [d] from the shifted sums; running_mean / running_var (fp32, or None) updated in place."""
        d = sums.numel() // 2
        mean = torch.empty(d, dtype=_F32, device=sums.device)
        rstd = torch.empty(d, dtype=_F32, device=sums.device)
        with torch.cuda.device(sums.device):
            _lib.call("sgf_bn_finalize", _ptr(sums), _ptr(shift), float(n_total), float(eps), float(momentum),
                      _ptr(running_mean), _ptr(running_var), d, _ptr(mean), _ptr(rstd), _stream(sums.device))
        return mean, rstd

    @staticmethod
    def bn_apply(x, mean, rstd, gamma, beta, res, relu: bool) -> torch.Tensor:
        n, d = x.shape
        dev = x.device
        y = torch.empty((n, d), dtype=x.dtype, device=dev)
        with torch.cuda.device(dev):
            _lib.call("sgf_bn_apply", _ptr(x), _ld(x), _ptr(mean), _ptr(rstd), _ptr(gamma), _ptr(beta),
                      _ptr(res), _ld(res), int(relu), n, d, _code(x), _ptr(y), y.stride(0),
                      _stream(dev))
        return y

    @staticmethod
    def bn_bwd_stats(gy, x, mean, rstd, gamma, beta, relu: bool) -> torch.Tensor:
        n, d = x.shape
        dev = x.device
        stats = torch.empty(2 * d, dtype=_F32, device=dev)
        ws = _workspace(dev, "col", _lib.load().sgf_colstats_workspace_bytes(n, d))
        with torch.cuda.device(dev):
            _lib.call("sgf_bn_bwd_stats", _ptr(gy), _ld(gy), _ptr(x), _ld(x), _ptr(mean), _ptr(rstd),
                      _ptr(gamma), _ptr(beta), int(relu), n, d, _code(x), _ptr(stats), _ptr(ws),
                      ws.numel(), _stream(dev))
        return stats

    @staticmethod
    def bn_bwd_stats2(gy, gy2, x, mean, rstd, gamma, beta, relu: bool) -> torch.Tensor:
        """bn_bwd_stats of the gradient gy + gy2 (two consumers of the BatchNorm's output; gy2 may be None)."""
        n, d = x.shape
        dev = x.device
        stats = torch.empty(2 * d, dtype=_F32, device=dev)
        ws = _workspace(dev, "col", _lib.load().sgf_colstats_workspace_bytes(n, d))
        with torch.cuda.device(dev):
            _lib.call("sgf_bn_bwd_stats2", _ptr(gy), _ld(gy), _ptr(gy2), _ld(gy2), _ptr(x), _ld(x), _ptr(mean), _ptr(rstd),
                      _ptr(gamma), _ptr(beta), int(relu), n, d, _code(x), _ptr(stats), _ptr(ws), ws.numel(), _stream(dev))
        return stats

    @staticmethod
    def gram_ln_bwd_supported(m: int, k: int, dtype) -> bool:
        return dtype == _BF16 and bool(_lib.load().sgf_gram_ln_bwd_supported(int(m), int(k), _lib.SGF_BF16))

    @staticmethod
    def gram_ln_bwd(g, xin, mean, rstd, gamma, beta, relu: bool, b):
        """(dl^T b [m, k], sum dl [m], dgamma [m], dbeta [m]) with dl = the LayerNorm's input gradient, never written."""
        n, m = xin.shape
        k = b.shape[1]
        dev = xin.device
        out = torch.empty((m, k), dtype=_F32, device=dev)
        cs = torch.empty(m, dtype=_F32, device=dev)
        dg = torch.empty(m, dtype=_F32, device=dev)
        db = torch.empty(m, dtype=_F32, device=dev)
        ws = _workspace(dev, "attn", _lib.load().sgf_gram_workspace_bytes(n, m, k))
        with torch.cuda.device(dev):
            _lib.call("sgf_gram_ln_bwd", _ptr(g), _ld(g), _ptr(xin), _ld(xin), _ptr(mean), _ptr(rstd), _ptr(gamma), _ptr(beta),
                      int(relu), m, _ptr(b), _ld(b), k, n, _code(xin), _ptr(out), out.stride(0), _ptr(cs), _ptr(dg), _ptr(db),
                      _ptr(ws), ws.numel(), _stream(dev))
        return out, cs, dg, db

    @staticmethod
    def gram_bn_bwd_supported(m: int, k: int, dtype) -> bool:
        return dtype == _BF16 and bool(_lib.load().sgf_gram_bn_bwd_supported(int(m), int(k), _lib.SGF_BF16))

    @staticmethod
    def gram_bn_bwd(gy, gy2, z, mean, rstd, gamma, beta, relu: bool, stats, inv_n: float, training: bool, b):
        """(dz^T b [m, k] fp32, colsum(dz) [m]) with dz = bn_bwd_apply(gy + gy2, z, ...) never written (sgf_gram_bn_bwd)."""
        n, m = z.shape
        k = b.shape[1]
        dev = z.device
        out = torch.empty((m, k), dtype=_F32, device=dev)
        cs = torch.empty(m, dtype=_F32, device=dev)
        ws = _workspace(dev, "attn", _lib.load().sgf_gram_workspace_bytes(n, m, k))
        with torch.cuda.device(dev):
            _lib.call("sgf_gram_bn_bwd", _ptr(gy), _ld(gy), _ptr(gy2), _ld(gy2), _ptr(z), _ld(z), _ptr(mean), _ptr(rstd),
                      _ptr(gamma), _ptr(beta), int(relu), _ptr(stats), float(inv_n), int(training), m, _ptr(b), _ld(b), k, n,
                      _code(z), _ptr(out), out.stride(0), _ptr(cs), _ptr(ws), ws.numel(), _stream(dev))
        return out, cs

    @staticmethod
    def gram2_bn_bwd_supported(g, z, b1, b2) -> bool:
        """sgf_gram2_bn_bwd serves these operands (bf16, sizes, 16-byte aligned rows)?"""
        n, m = z.shape
        ok = all(t.dtype == _BF16 and t.dim() == 2 and t.stride(1) == 1 and t.stride(0) % 8 == 0 and t.data_ptr() % 16 == 0
                 for t in (g, z, b1, b2))
        return (ok and b1.shape == b2.shape and g.shape == z.shape
                and bool(_lib.load().sgf_gram2_bn_bwd_supported(int(m), int(b1.shape[1]), int(n), _lib.SGF_BF16)))

    @staticmethod
    def gram2_bn_bwd(g, z, mean, rstd, gamma, beta, relu: bool, stats, inv_n: float, training: bool, b1, b2, out1, out2):
        """(dz, colsum(dz)); out1 = dz^T b1, out2 = dz^T b2 (fp32, may be column slices of one matrix): the BatchNorm backward
        and both weight-gradient blocks of a GraphConv layer from one pass over (g, z) (sgf_gram2_bn_bwd)."""
        n, m = z.shape
        k = b1.shape[1]
        dev = z.device
        dz = torch.empty((n, m), dtype=z.dtype, device=dev)
        cs = torch.empty(m, dtype=_F32, device=dev)
        ws = _workspace(dev, "attn", _lib.load().sgf_gram_workspace_bytes(n, m, k))
        with torch.cuda.device(dev):
            _lib.call("sgf_gram2_bn_bwd", _ptr(g), _ld(g), _ptr(z), _ld(z), _ptr(mean), _ptr(rstd), _ptr(gamma), _ptr(beta),
                      int(relu), _ptr(stats), float(inv_n), int(training), m, _ptr(b1), _ld(b1), _ptr(b2), _ld(b2), k, n,
                      _code(z), _ptr(dz), dz.stride(0), _ptr(out1), out1.stride(0), _ptr(out2), out2.stride(0), _ptr(cs),
                      _ptr(ws), ws.numel(), _stream(dev))
        return dz, cs

    @staticmethod
    def bn_bwd_apply(gy, x, mean, rstd, gamma, beta, relu: bool, stats, inv_n: float,
                     training: bool) -> torch.Tensor:
        n, d = x.shape
        dev = x.device
        dx = torch.empty_like(x)
        with torch.cuda.device(dev):
            _lib.call("sgf_bn_bwd_apply", _ptr(gy), _ld(gy), _ptr(x), _ld(x), _ptr(mean), _ptr(rstd),
                      _ptr(gamma), _ptr(beta), int(relu), _ptr(stats), float(inv_n), int(training), n,
                      d, _code(x), _ptr(dx), _ld(dx), _stream(dev))
        return dx

    @staticmethod
    def colsum(x) -> torch.Tensor:
        n, d = x.shape
        dev = x.device
        out = torch.empty(d, dtype=_F32, device=dev)
        ws = _workspace(dev, "colsum", _lib.load().sgf_colsum_workspace_bytes(n, d))
        with torch.cuda.device(dev):
            _lib.call("sgf_colsum", _ptr(x), _ld(x), n, d, _code(x), _ptr(out), _ptr(ws), ws.numel(),
                      _stream(dev))
        return out

    # ---- dropout (+ residual), mask recomputed from the seed ----
    @staticmethod
    def dropout(x, res, p: float, seed: int) -> torch.Tensor:
        n, d = x.shape
        y = torch.empty((n, d), dtype=x.dtype, device=x.device)
        with torch.cuda.device(x.device):
            _lib.call("sgf_dropout", _ptr(x), _ld(x), _ptr(res), _ld(res), float(p), int(seed), n, d,
                      _code(x), _ptr(y), y.stride(0), _stream(x.device))
        return y

    # ---- N4: log_softmax + NLL on the training rows ----
    @staticmethod
    def nll_fwd(logits, labels, idx) -> torch.Tensor:
        n, c = logits.shape
        dev = logits.device
        out = torch.empty(1, dtype=_F32, device=dev)
        ws = _workspace(dev, "nll", _lib.load().sgf_nll_workspace_bytes(idx.numel()))
        with torch.cuda.device(dev):
            _lib.call("sgf_nll_fwd", _ptr(logits), logits.stride(0), n, c, _code(logits), _ptr(labels),
                      _ptr(idx), idx.numel(), _ptr(out), _ptr(ws), ws.numel(), _stream(dev))
        return out

    @staticmethod
    def nll_bwd(logits, labels, idx, gout, inv_denom: float) -> torch.Tensor:
        n, c = logits.shape
        dev = logits.device
        d = torch.empty((n, c), dtype=logits.dtype, device=dev)
        with torch.cuda.device(dev):
            _lib.call("sgf_nll_bwd", _ptr(logits), logits.stride(0), n, c, _code(logits), _ptr(labels),
                      _ptr(idx), idx.numel(), _ptr(gout), float(inv_denom), _ptr(d), d.stride(0),
                      _stream(dev))
        return d

    @staticmethod
    def sum_n(xs) -> torch.Tensor:
        """sum of up to 8 equally shaped [n, d] tensors in one pass."""
        xs = [_rows(x) for x in xs]
        n, d = xs[0].shape
        y = torch.empty((n, d), dtype=xs[0].dtype, device=xs[0].device)
        k = len(xs)
        ptrs = (ctypes.c_void_p * k)(*[x.data_ptr() for x in xs])
        lds = (ctypes.c_int64 * k)(*[x.stride(0) for x in xs])
        with torch.cuda.device(y.device):
            _lib.call("sgf_sum_n", ptrs, lds, k, n, d, _code(y), _ptr(y), y.stride(0), _stream(y.device))
        return y

    # ---- T7 fused: logits = (a x1 + b x2) W^T + bias ----
    @staticmethod
    def combine_fc_supported(d: int, classes: int, dtype) -> bool:
        """bf16: d % 32 == 0, d <= 256, classes <= 64 (csrc/head.hip); fp32: d % 4 == 0, d <= 256 and the class count
        PADDED to a multiple of 4 (ops.combine_fc pads W / bias with zero rows) up to 256 (csrc/linear_f32.hip)."""
        if dtype == _F32:
            return bool(_lib.load().sgf_combine_fc_supported(d, (classes + 3) // 4 * 4, _lib.SGF_F32))
        if dtype == _BF16 and classes > 64:      # the same exact-fp32 kernel with bf16 rows on the wire (C = 172: papers100M)
            return bool(_lib.load().sgf_combine_fc_supported(d, (classes + 3) // 4 * 4, _lib.SGF_BF16))
        return dtype == _BF16 and bool(_lib.load().sgf_combine_fc_supported(d, classes, _lib.SGF_BF16))

    @staticmethod
    def combine_fc_mapped_supported(d: int, classes: int, dtype) -> bool:
        """the row-mapped forms (sgf_combine_fc_*_mapped): the bf16 kernels of csrc/head.hip only"""
        return dtype == _BF16 and classes <= 64 and bool(_lib.load().sgf_combine_fc_supported(d, classes, _lib.SGF_BF16))

    @staticmethod
    def combine_fc_fwd(x1, a: float, x2, b: float, w, bias, row_map=None) -> torch.Tensor:
        """row_map (int32 permutation): row j of the product is stored as row row_map[j]"""
        n, d = x1.shape
        c = w.shape[0]
        HipKernels._check_row_map(row_map, n, x1.device)
        logits = torch.empty((n, c), dtype=_F32, device=x1.device)
        with torch.cuda.device(x1.device):
            if row_map is None:
                _lib.call("sgf_combine_fc_fwd", _ptr(x1), _ld(x1), float(a), _ptr(x2), _ld(x2), float(b), _ptr(w),
                          _ptr(bias), n, d, c, _code(x1), _ptr(logits), logits.stride(0), _stream(x1.device))
            else:
                _lib.call("sgf_combine_fc_fwd_mapped", _ptr(x1), _ld(x1), float(a), _ptr(x2), _ld(x2), float(b), _ptr(w),
                          _ptr(bias), n, d, c, _code(x1), _ptr(logits), logits.stride(0), _ptr(row_map), _stream(x1.device))
        return logits

    @staticmethod
    def combine_fc_bwd(g, w, a: float, b: float, dtype, row_map=None):
        """row_map: row j of dx1 / dx2 comes from row row_map[j] of g"""
        n, c = g.shape
        d = w.shape[1]
        HipKernels._check_row_map(row_map, n, g.device)
        dx1 = torch.empty((n, d), dtype=dtype, device=g.device)
        dx2 = torch.empty((n, d), dtype=dtype, device=g.device)
        with torch.cuda.device(g.device):
            if row_map is None:
                _lib.call("sgf_combine_fc_bwd", _ptr(g), g.stride(0), _ptr(w), n, d, c, float(a), float(b),
                          _lib.SGF_BF16 if dtype == _BF16 else _lib.SGF_F32, _ptr(dx1), dx1.stride(0), _ptr(dx2),
                          dx2.stride(0), _stream(g.device))
            else:
                _lib.call("sgf_combine_fc_bwd_mapped", _ptr(g), g.stride(0), _ptr(w), n, d, c, float(a), float(b),
                          _lib.SGF_BF16, _ptr(dx1), dx1.stride(0), _ptr(dx2), dx2.stride(0), _ptr(row_map),
                          _stream(g.device))
        return dx1, dx2

    @staticmethod
    def combine_fc_bwd_g(g, w, a: float, b: float, row_map=None):
        """(dx1, dx2, gp): combine_fc_bwd for bf16 storage and at most 64 classes that also returns gp = the logits' gradient
        in bf16, [n, 16 ceil(c / 16)], zero-padded, in the module's row order (sgf_combine_fc_bwd_g)."""
        n, c = g.shape
        d = w.shape[1]
        HipKernels._check_row_map(row_map, n, g.device)
        dx1 = torch.empty((n, d), dtype=_BF16, device=g.device)
        dx2 = torch.empty((n, d), dtype=_BF16, device=g.device)
        gp = torch.empty((n, (c + 15) // 16 * 16), dtype=_BF16, device=g.device)
        with torch.cuda.device(g.device):
            _lib.call("sgf_combine_fc_bwd_g", _ptr(g), g.stride(0), _ptr(w), n, d, c, float(a), float(b), _lib.SGF_BF16, _ptr(dx1),
                      dx1.stride(0), _ptr(dx2), dx2.stride(0), _ptr(row_map), _ptr(gp), gp.stride(0), _stream(g.device))
        return dx1, dx2, gp

    @staticmethod
    def _check_row_map(row_map, n: int, device):
        """The *_mapped kernels read `const int32_t*`: anything else would scatter rows out of bounds."""
        if row_map is not None and not (row_map.dtype == torch.int32 and row_map.is_contiguous() and row_map.numel() == n
                                        and row_map.device == device):
            raise RuntimeError(f"combine_fc: row_map must be a contiguous int32 permutation of {n} rows on {device}, got "
                               f"{row_map.dtype} x {tuple(row_map.shape)} on {row_map.device}")

    # ---- T6 / K8: Linear (+ BatchNorm statistics) as one streaming pass ----
    @staticmethod
    def gcn_epilogue_supported(d_in: int, d_out: int, dtype) -> bool:
        """bf16 storage: square layers of 64 / 128 / 256 (csrc/rowgemm.hip); fp32 storage: any widths % 4 == 0 up to
        256 (csrc/linear_f32.hip, exact-fp32 matrix cores)."""
        if dtype not in (_BF16, _F32):
            return False
        return bool(_lib.load().sgf_gcn_epilogue_supported(d_in, d_out, _lib.SGF_BF16 if dtype == _BF16 else _lib.SGF_F32))

    @staticmethod
    def gcn_epilogue_stats(a, w, bias, shift=None, want_stats=False):
        """y = a w^T + bias; with want_stats also [sum(y - shift) | sum((y - shift)^2)] per column of the
        rounded y.  w in a's dtype [d_out, d_in], bias / shift fp32."""
        n, d_in = a.shape
        d_out = w.shape[0]
        dev = a.device
        y = torch.empty((n, d_out), dtype=a.dtype, device=dev)
        stats = torch.empty(2 * d_out, dtype=_F32, device=dev) if want_stats else None
        lib = _lib.load()
        ws = _workspace(dev, "gcn_epi", lib.sgf_gcn_epilogue_workspace_bytes(n, d_out)) if want_stats else None
        with torch.cuda.device(dev):
            _lib.call("sgf_gcn_epilogue_stats", _ptr(a), _ld(a), _ptr(w), w.stride(0), _ptr(bias), n, d_in, d_out,
                      _code(a), _ptr(y), _ld(y), _ptr(shift), _ptr(stats), _ptr(ws),
                      0 if ws is None else ws.numel(), _stream(dev))
        return y, stats

    @staticmethod
    def gcn_epilogue_cat(a1, a2, w, bias, shift=None, want_stats=False):
        """y = [a1 | a2] w^T + bias (w [d, 2 d]).  bf16, square blocks: ONE pass over a1 and a2 (sgf_gcn_epilogue_cat: W
        resident in LDS for d <= 128, a paired launch for d = 256; SGF_GCN_CAT=0 keeps the two-pass form).  Otherwise two
        streaming passes: a1's product stays in the matrix cores' accumulator layout (an opaque scratch buffer) and is
        added in a2's pass."""
        n, d1 = a1.shape
        d2, d = a2.shape[1], w.shape[0]
        dev = a1.device
        lib = _lib.load()
        if (a1.dtype == _BF16 and d1 == d2 == d and _one_pass_cat() and lib.sgf_gcn_epilogue_cat_supported(d, _lib.SGF_BF16)
                and w.stride(1) == 1 and w.shape[1] == 2 * d):
            y = torch.empty((n, d), dtype=a1.dtype, device=dev)
            stats = torch.empty(2 * d, dtype=_F32, device=dev) if want_stats else None
            ws = _workspace(dev, "gcn_epi", lib.sgf_gcn_epilogue_workspace_bytes(n, d)) if want_stats else None
            with torch.cuda.device(dev):
                _lib.call("sgf_gcn_epilogue_cat", _ptr(a1), _ld(a1), _ptr(a2), _ld(a2), _ptr(w), w.stride(0), _ptr(bias), n,
                          d, _code(a1), _ptr(y), _ld(y), _ptr(shift), _ptr(stats), _ptr(ws),
                          0 if ws is None else ws.numel(), _stream(dev))
            return y, stats
        part = _workspace(dev, "gcn_part", lib.sgf_gcn_epilogue_dtype_partial_bytes(n, d, _code(a1)))
        y = torch.empty((n, d), dtype=a1.dtype, device=dev)
        stats = torch.empty(2 * d, dtype=_F32, device=dev) if want_stats else None
        ws = _workspace(dev, "gcn_epi", lib.sgf_gcn_epilogue_workspace_bytes(n, d)) if want_stats else None
        w1, w2 = w[:, :d1], w[:, d1:]
        with torch.cuda.device(dev):
            _lib.call("sgf_gcn_epilogue_partial", _ptr(a1), _ld(a1), _ptr(w1), w.stride(0), _ptr(bias), n, d1, d,
                      _code(a1), _ptr(part), part.numel(), _stream(dev))
            _lib.call("sgf_gcn_epilogue_stats_add", _ptr(a2), _ld(a2), _ptr(w2), w.stride(0), _ptr(part),
                      part.numel(), n, d2, d, _code(a2), _ptr(y), _ld(y), _ptr(shift), _ptr(stats), _ptr(ws),
                      0 if ws is None else ws.numel(), _stream(dev))
        return y, stats

    @staticmethod
    def stem_pair_supported(d_in: int, d_out: int, dtype) -> bool:
        return dtype == _BF16 and bool(_lib.load().sgf_stem_pair_supported(d_in, d_out, _lib.SGF_BF16))

    @staticmethod
    def stem_pair(x, w0, b0, w1, b1, shift0=None, want_stats0=False):
        """y0 = x w0^T + b0 (+ its BatchNorm column sums), y1 = x w1^T + b1 from one read of x (w1 None: y1 None)."""
        n, d_in = x.shape
        d_out = w0.shape[0]
        dev = x.device
        y0 = torch.empty((n, d_out), dtype=x.dtype, device=dev)
        y1 = torch.empty((n, d_out), dtype=x.dtype, device=dev) if w1 is not None else None
        stats = torch.empty(2 * d_out, dtype=_F32, device=dev) if want_stats0 else None
        lib = _lib.load()
        ws = _workspace(dev, "gcn_epi", lib.sgf_gcn_epilogue_workspace_bytes(n, d_out)) if want_stats0 else None
        with torch.cuda.device(dev):
            _lib.call("sgf_stem_pair", _ptr(x), _ld(x), n, d_in, _ptr(w0), w0.stride(0), _ptr(b0), _ptr(w1),
                      0 if w1 is None else w1.stride(0), _ptr(b1), d_out, _code(x), _ptr(y0), _ld(y0), _ptr(y1),
                      0 if y1 is None else _ld(y1), _ptr(shift0), _ptr(stats), _ptr(ws), 0 if ws is None else ws.numel(),
                      _stream(dev))
        return y0, y1, stats

    @staticmethod
    def gcn_epilogue_dx(dy, w):
        """dx = dy w  (w [d_out, d_in] in dy's dtype; may be a column slice of a wider matrix)."""
        n, d_out = dy.shape
        d_in = w.shape[1]
        dx = torch.empty((n, d_in), dtype=dy.dtype, device=dy.device)
        with torch.cuda.device(dy.device):
            _lib.call("sgf_gcn_epilogue_dx", _ptr(dy), _ld(dy), _ptr(w), w.stride(0), n, d_in, d_out, _code(dy),
                      _ptr(dx), _ld(dx), _stream(dy.device))
        return dx

    @staticmethod
    def gcn_epilogue_dx2_acc_supported(d: int, dtype) -> bool:
        return dtype == _BF16 and bool(_lib.load().sgf_gcn_epilogue_dx2_acc_supported(int(d), _lib.SGF_BF16))

    @staticmethod
    def gcn_epilogue_dx2_acc(dz, w, gadd, acc_in):
        """(dz W[:, :d], dz W[:, d:] + gadd + acc_in): both input gradients of the two-operand Linear, the second one added to
        the running gradient of x0 (sgf_gcn_epilogue_dx2_acc); gadd / acc_in may be None."""
        n, d = dz.shape
        dy = torch.empty_like(dz)
        acc = torch.empty_like(dz)
        with torch.cuda.device(dz.device):
            _lib.call("sgf_gcn_epilogue_dx2_acc", _ptr(dz), _ld(dz), _ptr(w), _ld(w), n, d, _code(dz), _ptr(dy), _ld(dy),
                      _ptr(gadd), _ld(gadd), _ptr(acc_in), _ld(acc_in), _ptr(acc), _ld(acc), _stream(dz.device))
        return dy, acc

    @staticmethod
    def gcn_bn_bwd_dx_supported(d: int, dtype) -> bool:
        return dtype == _BF16 and bool(_lib.load().sgf_gcn_bn_bwd_dx_supported(int(d), _lib.SGF_BF16))

    @staticmethod
    def gcn_bn_bwd_dx(gy, z, mean, rstd, gamma, beta, relu: bool, stats, inv_n: float, training: bool, w, acc_in,
                      last: bool, add_gy: bool):
        """(dz, dy, acc): BatchNorm backward + both input gradients of W [Ax | x0] in one launch (sgf_gcn_bn_bwd_dx).
        `acc_in`: the running gradient of x0 from the layers processed so far (opaque uint8 tensor) or None; `acc`: the
        new running sum (opaque) or, with last=True, the total as a row-major [n, d] tensor."""
        n, d = z.shape
        dev = z.device
        dz = torch.empty((n, d), dtype=z.dtype, device=dev)
        dy = torch.empty((n, d), dtype=z.dtype, device=dev)
        nbytes = _lib.load().sgf_gcn_epilogue_partial_bytes(n, d)
        acc_out = None if last else torch.empty(max(nbytes, 16), dtype=torch.uint8, device=dev)
        dx0 = torch.empty((n, d), dtype=z.dtype, device=dev) if last else None
        ws = _workspace(dev, "gcn_bwd_sync", _lib.load().sgf_gcn_bn_bwd_dx_workspace_bytes(n, d))
        with torch.cuda.device(dev):
            _lib.call("sgf_gcn_bn_bwd_dx", _ptr(gy), _ld(gy), _ptr(z), _ld(z), _ptr(mean), _ptr(rstd), _ptr(gamma), _ptr(beta),
                      int(relu), _ptr(stats), float(inv_n), int(training), _ptr(w), w.stride(0), n, d, _code(z), _ptr(dz),
                      _ld(dz), _ptr(dy), _ld(dy), _ptr(acc_in), _ptr(acc_out), nbytes, _ptr(dx0), _ld(dx0), int(add_gy),
                      _ptr(ws), ws.numel(), _stream(dev))
        return dz, dy, (dx0 if last else acc_out)

    @staticmethod
    def gcn_epilogue_dx2(dy, w1, w2, pair: bool = True):
        """(dy w1, dy w2) for two [d, d] column blocks of one weight matrix, dy read from HBM once (paired launch)."""
        n, d = dy.shape
        dx1 = torch.empty((n, d), dtype=dy.dtype, device=dy.device)
        dx2 = torch.empty((n, d), dtype=dy.dtype, device=dy.device)
        assert w1.stride(0) == w2.stride(0)
        with torch.cuda.device(dy.device):
            _lib.call("sgf_gcn_epilogue_dx2", _ptr(dy), _ld(dy), _ptr(w1), _ptr(w2), w1.stride(0), n, d, _code(dy),
                      _ptr(dx1), _ld(dx1), _ptr(dx2), _ld(dx2), int(pair), _stream(dy.device))
        return dx1, dx2

    # ---- T7 ----
    @staticmethod
    def axpby(x1, a: float, x2, b: float) -> torch.Tensor:
        n, d = x1.shape
        y = torch.empty((n, d), dtype=x1.dtype, device=x1.device)
        with torch.cuda.device(x1.device):
            _lib.call("sgf_axpby", _ptr(x1), _ld(x1), float(a), _ptr(x2), _ld(x2), float(b), n, d,
                      _code(x1), _ptr(y), y.stride(0), _stream(x1.device))
        return y
