"""bench.py: the SpMM roofline (HIP-event timing of every SpMM launch on the launch stream, PMC traffic table) and the
whole-step HBM fraction of SURVEY.md §8d."""
from __future__ import annotations

import os
import sys
import time

import torch
import json

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from sgformer_amd import ops, synth  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s measured copy ceiling)

# HBM bytes per SpMM launch from rocprofv3 PMC passes (separate --pmc FETCH_SIZE / WRITE_SIZE runs of
# scripts/spmm_pmc_target.py; FETCH_SIZE x2 per the gfx950 correction of MI355X_MICROARCH.md §HBM).  PMC
# counters cannot be collected from inside this process, so the figures live in a tracked file written
# from those passes, keyed on graph kind / dtype / kernel.
PMC_FILE = os.path.join(ROOT, "profiles", "r06_spmm_pmc.json")
SPMM_SOURCES = ("spmm.hip", "spmm_tile.hip", "spmm_pack.hip", "spmm_plan.hip", "spmm_shared.h")


def spmm_source_sha16() -> str:
    """sha256 (first 16 hex digits) of the SpMM sources the PMC passes were taken with."""
    import hashlib
    h = hashlib.sha256()
    for name in SPMM_SOURCES:
        with open(os.path.join(ROOT, "sgformer_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def pmc_traffic(graph_kind: str, dtype: str, kernel: str, reordered: bool):
    """(HBM bytes per launch, source) — or (None, why) when no pass exists or the passes are STALE: the file records
    the hash of the kernel sources it was measured with, and a number measured on other code is not reported."""
    try:
        table = json.load(open(PMC_FILE))
    except (OSError, ValueError):
        return None, None
    if table.get("_source_sha16") != spmm_source_sha16():
        return None, "profiles/r06_spmm_pmc.json is older than csrc/spmm*.hip: re-run scripts/pmc_passes.sh"
    tag = "reordered" if reordered else "given"
    e = table.get(f"{graph_kind}/{dtype}/{kernel}/{tag}")
    if not e:
        return None, None
    total = e["hbm_bytes_per_launch"]
    if kernel in ("k_spmm_row", "k_spmm_seg_bf16x2"):
        # one sgf_spmm call = the row kernel + the long-row path (hub rows: k_spmm_long_seg / _fin), timed together by the
        # HIP events above — so their traffic is reported together too (R-MAT: 29.8 + 24.1 + 0.1 GB)
        for extra in ("k_spmm_long_seg", "k_spmm_long_fin"):
            x = table.get(f"{graph_kind}/{dtype}/{extra}/{tag}") or table.get(f"{graph_kind}/{dtype}/{extra}/given")
            if x:
                total += x["hbm_bytes_per_launch"]
    return total, "profiles/r06_spmm_pmc.json"


class SpmmTimer:
    """HIP-event timing of every SpMM launch on the launch stream (torch's current stream)."""

    def __init__(self):
        self.pairs, self.bytes_alg, self.bytes_gather, self.active, self.kernels = [], [], [], False, []
        self._orig = (ops.K.spmm, getattr(ops.K, "spmm_blocked", None), getattr(ops.K, "spmm_tile", None))

    def _wrap(self, orig, blocked):
        timer = self

        def timed(rowptr, b, *rest, **kw):
            # K.spmm(rowptr, colind, val, x, n_rows, ...) / K.spmm_blocked(rowptr, plan, x, n_rows, ...)
            if not timer.active or torch.cuda.is_current_stream_capturing():
                return orig(rowptr, b, *rest, **kw)
            x, n_rows = (rest[0], rest[1]) if blocked else (rest[1], rest[2])
            nnz = int(b.nnz) if blocked else b.numel()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            y = orig(rowptr, b, *rest, **kw)
            e1.record()
            s, d = x.element_size(), x.shape[1]
            bf16 = x.dtype == torch.bfloat16
            # the library's dispatch (csrc/spmm.hip::launch): stream kernel for re-ordered graphs and for bf16 rows of
            # 65-128 elements, wave per row above 128, sub-wave per row below
            timer.kernels.append("k_spmm_blk" if blocked else (
                "k_spmm_seg_bf16x2" if bf16 and d % 8 == 0 and (kw.get("stream_hint") and d > 128 or 64 < d <= 128)
                else ("k_spmm_sub" if d <= 128 else "k_spmm_row")))
            timer.pairs.append((e0, e1))
            timer.bytes_alg.append(nnz * 8 + (n_rows + 1) * 8 + x.shape[0] * d * s + n_rows * d * s)
            timer.bytes_gather.append(nnz * (8 + d * s) + (n_rows + 1) * 8 + n_rows * d * s)
            return y

        return timed

    def _wrap_tile(self, orig):
        timer = self

        def timed(plan, x, n_rows, *rest, **kw):
            if not timer.active or torch.cuda.is_current_stream_capturing():
                return orig(plan, x, n_rows, *rest, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            y = orig(plan, x, n_rows, *rest, **kw)
            e1.record()
            s, d, nnz = x.element_size(), x.shape[1], int(plan.nnz)
            timer.kernels.append("k_spmm_tile_bf16")
            timer.pairs.append((e0, e1))
            timer.bytes_alg.append(nnz * 8 + (n_rows + 1) * 8 + x.shape[0] * d * s + n_rows * d * s)
            timer.bytes_gather.append(nnz * (8 + d * s) + (n_rows + 1) * 8 + n_rows * d * s)
            return y

        return timed

    def install(self):
        ops.K.spmm = self._wrap(self._orig[0], False)
        ops.K.spmm_blocked = self._wrap(self._orig[1], True)
        ops.K.spmm_tile = self._wrap_tile(self._orig[2])

    def uninstall(self):
        ops.K.spmm, ops.K.spmm_blocked, ops.K.spmm_tile = self._orig

    def reset(self):
        self.pairs, self.bytes_alg, self.bytes_gather, self.kernels = [], [], [], []

    def summary(self):
        if not self.pairs:
            return None
        ms = [a.elapsed_time(b) for a, b in self.pairs]
        mean_ms = sum(ms) / len(ms)
        alg = sum(self.bytes_alg) / len(self.bytes_alg)
        gat = sum(self.bytes_gather) / len(self.bytes_gather)
        achieved = alg / (mean_ms * 1e-3) / 1e9
        kern = max(set(self.kernels), key=self.kernels.count)
        entry = {"k_spmm_blk": "sgf_spmm_blocked", "k_spmm_seg_bf16x2": "sgf_spmm_stream",
                 "k_spmm_tile_bf16": "sgf_spmm_tile"}.get(kern, "sgf_spmm")
        return {"kernel": f"{kern} ({entry})", "bound": "hbm", "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": None, "launches": len(ms), "mean_launch_ms": round(mean_ms, 4),
                "algorithmic_bytes": int(alg), "gather_bytes": int(gat),
                "gather_GBps": round(gat / (mean_ms * 1e-3) / 1e9, 1)}


def step_compulsory_bytes(n: int, nnz: int, f: int, d: int, lg: int, lt: int, s: int) -> int:
    """SURVEY.md §8d, 'whole-step compulsory bytes per node (fully fused ideal)':
        s [2 f + d (3 + 2 Lg + 3 Lt)] (1 fwd + 2 bwd) + 8 kbar 2 Lg
    times N (kbar = nnz / N): every [N, d] tensor the recipe cannot avoid read / written once per direction, the CSR's 8 bytes
    per stored entry once per SpMM launch.  ogbn-products shape, bf16: 22 104 B per node = 54.1 GB per step."""
    return int(n * s * (2 * f + d * (3 + 2 * lg + 3 * lt)) * 3 + 8 * nnz * 2 * lg)


def step_roofline(n: int, nnz: int, f: int, d: int, cfg: dict, dtype: str, ms_per_step: float):
    """`roofline.step`: the end-to-end HBM fraction of one training step = compulsory bytes / step time / 8 TB/s."""
    lg, lt = int(cfg.get("gnn_num_layers", 0)), int(cfg.get("trans_num_layers", cfg.get("num_layers", 1)))
    s = 4 if dtype == "f32" else 2
    b = step_compulsory_bytes(n, nnz, f, d, lg, lt, s)
    gbs = b / (ms_per_step * 1e-3) / 1e9
    return {"compulsory_bytes": b, "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(gbs / HBM_PEAK_GBS, 4),
            "formula": "SURVEY.md 8d: N s [2 f + d (3 + 2 Lg + 3 Lt)] (1 fwd + 2 bwd) + 8 nnz 2 Lg"}
