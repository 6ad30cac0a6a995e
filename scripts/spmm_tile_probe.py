#!/usr/bin/env python
"""sgf_spmm_tile on ONE box against the stream kernel (k_spmm_seg_bf16x2) on the same re-ordered CSR:
plan statistics, one-off costs, launch times (interleaved, median), ablations (SGF_SPMM_TILE_DEBUG).

    python scripts/spmm_tile_probe.py [--graph community|powerlaw|uniform] [--n 2449029] [--deg 50.5] [--d 256]
                                      [--params cap,min_count,max_rows ...] [--ablate]
Prints one JSON line per measurement."""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sgformer_amd import _lib, ops, synth  # noqa: E402


def timed(fn, reps=15, warm=12):
    """median of `reps` launches after `warm` untimed ones (the first ~20 ms after an idle phase run at ramping clocks:
    with a single warm-up launch the same kernel measured 7 % slower)"""
    ts = []
    for i in range(reps + warm):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        if i >= warm:
            ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--graph", default="community")
    ap.add_argument("--n", type=int, default=2449029)
    ap.add_argument("--deg", type=float, default=50.5)
    ap.add_argument("--d", type=int, default=256)
    ap.add_argument("--params", nargs="*", default=["512,2,128"])
    ap.add_argument("--ablate", action="store_true")
    ap.add_argument("--sweep", action="store_true")
    ap.add_argument("--fixed-blocks", action="store_true", help="blocks of exactly max_rows rows (no community alignment)")
    ap.add_argument("--order", default="reorder", choices=["reorder", "planted"],
                    help="planted: the generator's own numbering and community labels (community graph only) — "
                         "the bound a perfect sgf_reorder would reach")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    n = a.n
    gen = {"community": synth.synthetic_graph_community, "uniform": synth.synthetic_graph,
           "powerlaw": getattr(synth, "synthetic_graph_community_powerlaw", None)}[a.graph]
    t0 = time.time()
    if a.order == "planted":
        ei, lab = gen(n, a.deg, seed=123, device=dev, shuffle_ids=False, return_labels=True)
        cs, comm = lab.int().contiguous(), lab
        g = ops.CSRGraph(ei, n, validate=False)
    else:
        ei = gen(n, a.deg, seed=123, device=dev)
        torch.cuda.synchronize()
        t0 = time.time()
        perm, inv, comm = ops.K.reorder(ei, n, *ops.REORDER_ITERS)
        torch.cuda.synchronize()
        cs = comm[perm.long()].contiguous()
        g = ops.CSRGraph(inv.long()[ei], n, validate=False)
    t_reorder = time.time() - t0
    del ei
    nnz = g.nnz
    x = torch.randn(n, a.d, device=dev).to(torch.bfloat16)
    alg = nnz * 8 + (n + 1) * 8 + 2 * n * a.d * 2
    out = {"graph": a.graph, "n": n, "nnz": nnz, "d": a.d, "order": a.order, "algorithmic_bytes": alg, "reorder_s": round(t_reorder, 3),
           "communities": int(comm.max()) + 1}
    print(json.dumps(out), flush=True)

    def report(name, ms, **kw):
        print(json.dumps({"kernel": name, "ms": round(ms, 3), "frac_of_8TBps": round(alg / (ms * 1e-3) / 8e12, 4), **kw}),
              flush=True)

    os.environ["SGF_SPMM_KERNEL"] = ""
    t_stream = timed(lambda: ops.K.spmm(g.rowptr, g.colind, g.val, x, n, long_segments=g.long_segments, stream_hint=True))
    report("k_spmm_seg_bf16x2 (stream)", t_stream)
    y_ref = ops.K.spmm(g.rowptr, g.colind, g.val, x, n, long_segments=g.long_segments, stream_hint=True).float()
    for prm in a.params:
        cap, mc, mr = (int(t) for t in prm.split(","))
        torch.cuda.synchronize()
        t0 = time.time()
        g.blk_row = ops.K.tile_blocks(None if a.fixed_blocks else cs, n, mr, dev)
        plan = ops.TilePlan(g.rowptr, g.colind, g.val, n, g.blk_row, cap=cap, min_count=mc)
        torch.cuda.synchronize()
        t_plan = time.time() - t0
        y = ops.K.spmm_tile(plan, x, n)
        rel = float((y.float() - y_ref).norm() / y_ref.norm())
        ms = timed(lambda: ops.K.spmm_tile(plan, x, n))
        report(f"k_spmm_tile cap={cap} min_count={mc} max_rows={mr}", ms, blocks=plan.nb,
               tile_fraction=round(plan.tile_fraction, 4), tile_density=round(plan.tile_density, 4),
               staged_rows_per_node=round(plan.staged_rows / n, 3), fragments=plan.fragments,
               tile_bytes=plan.tile_bytes, dense_tile_bytes=plan.fragments * 2048, rem_entries=plan.rem_entries, long_segments=plan.long_segments,
               plan_s=round(t_plan, 3), rel_diff_vs_stream=rel)
        if a.sweep:
            # XCD chunk: how many consecutive row blocks one XCD takes before the next stripe (SGF_SPMM_TILE_CHUNK);
            # two interleaved rounds so that clock drift shows
            for rnd in range(2):
                for ch in (0, 8, 16, 32, 64, 128, 256, 512, 1024, 4096):
                    if ch:
                        os.environ["SGF_SPMM_TILE_CHUNK"] = str(ch); _lib.load().sgf_reload_env()
                    else:
                        os.environ.pop("SGF_SPMM_TILE_CHUNK", None); _lib.load().sgf_reload_env()
                    report(f"  XCD chunk {ch or 'default'} blocks (round {rnd})", timed(lambda: ops.K.spmm_tile(plan, x, n)))
            os.environ.pop("SGF_SPMM_TILE_CHUNK", None); _lib.load().sgf_reload_env()
        if a.ablate:
            for dbg, what in ((1, "no tile phase (gathers + stores only)"), (2, "no gathers (tiles + stores only)"),
                              (3, "neither (skeleton)"), (16, "gathers clamped to 4096 rows (all L2 hits)"),
                              (17, "no tile phase + gathers clamped (L2 hits)"), (32, "no multiply-adds in the gather loop"),
                              (33, "no tile phase, no multiply-adds"), (49, "no tile phase, L2-hit gathers, no multiply-adds"),
                              (128, "tile phase without matrix-core work"), (130, "no gathers, tile phase without MFMA"),
                              ):
                os.environ["SGF_SPMM_TILE_DEBUG"] = str(dbg); _lib.load().sgf_reload_env()
                report(f"  ablation [{what}]", timed(lambda: ops.K.spmm_tile(plan, x, n)))
            os.environ.pop("SGF_SPMM_TILE_DEBUG"); _lib.load().sgf_reload_env()
        del plan


if __name__ == "__main__":
    main()
