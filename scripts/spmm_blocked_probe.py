#!/usr/bin/env python
"""SpMM probe: plain wave-per-row kernel vs re-ordered CSR vs LDS-staged row-block kernel.

    python scripts/spmm_blocked_probe.py --graph community --n 2449029 --deg 50.5 --dtype bf16

Prints one JSON line per measurement: one-off costs (CSR build, sgf_reorder, sgf_spmm_plan), the
plan's statistics, and the SpMM launch time / roofline fraction on ALGORITHMIC bytes
(nnz*8 + (N+1)*8 + 2*N*d*s) for each variant and block shape.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sgformer_amd import ops, synth  # noqa: E402


def timed(fn, reps=5, warm=2):
    for _ in range(warm):
        fn()
    evs = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in evs)
    return ms[len(ms) // 2]


def wall(fn):
    torch.cuda.synchronize()
    t = time.perf_counter()
    r = fn()
    torch.cuda.synchronize()
    return r, (time.perf_counter() - t) * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--graph", default="community", choices=["community", "uniform", "local", "skewed"])
    ap.add_argument("--n", type=int, default=2449029)
    ap.add_argument("--deg", type=float, default=50.5)
    ap.add_argument("--d", type=int, default=256)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--shapes", default="128x288,64x144,128x128,64x64,32x72")
    ap.add_argument("--seed", type=int, default=123)
    ap.add_argument("--ablate", action="store_true", help="time the row-block kernel with parts switched off")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    dtype = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    gen = {"community": synth.synthetic_graph_community, "uniform": synth.synthetic_graph,
           "local": synth.synthetic_graph_local, "skewed": synth.synthetic_graph_skewed}[a.graph]
    t0 = time.perf_counter()
    ei = gen(a.n, a.deg, seed=a.seed, device=dev)
    n, nnz, d = a.n, int(ei.shape[1]), a.d
    s = 2 if dtype == torch.bfloat16 else 4
    alg = nnz * 8 + (n + 1) * 8 + 2 * n * d * s
    print(json.dumps({"graph": a.graph, "n": n, "nnz": nnz, "d": d, "dtype": a.dtype, "algorithmic_bytes": alg,
                      "gen_s": round(time.perf_counter() - t0, 1)}), flush=True)
    x = torch.randn(n, d, device=dev).to(dtype)

    def report(tag, ms, **kw):
        print(json.dumps({"variant": tag, "launch_ms": round(ms, 4), "GBps_alg": round(alg / ms / 1e6, 1),
                          "frac_of_8TBps": round(alg / ms / 1e6 / 8000, 4), **kw}), flush=True)

    g, t_csr = wall(lambda: ops.CSRGraph(ei, n, validate=False))
    report("plain kernel, given node order", timed(lambda: ops.K.spmm(g.rowptr, g.colind, g.val, x, n,
                                                                        long_segments=g.long_segments)),
           csr_build_ms=round(t_csr, 1))
    (perm, inv, comm), t_re = wall(lambda: ops.K.reorder(ei, n, *ops.REORDER_ITERS))
    ncomm = int(comm.max()) + 1
    g2, t_csr2 = wall(lambda: ops.CSRGraph(inv.long()[ei], n, validate=False))
    del ei
    xp = ops.gather_rows(x, perm)
    report("plain kernel, sgf_reorder order", timed(lambda: ops.K.spmm(g2.rowptr, g2.colind, g2.val, xp, n,
                                                                         long_segments=g2.long_segments)),
           reorder_ms=round(t_re, 1), communities=ncomm, relabel_csr_ms=round(t_csr2, 1))
    if a.ablate:
        os.environ["SGF_SPMM_ROW_UNROLL"] = "16"
        report("plain kernel, sgf_reorder order, 16 gathers in flight", timed(lambda: ops.K.spmm(
            g2.rowptr, g2.colind, g2.val, xp, n, long_segments=g2.long_segments)))
        del os.environ["SGF_SPMM_ROW_UNROLL"]
    y_ref = ops.K.spmm(g2.rowptr, g2.colind, g2.val, xp, n, long_segments=g2.long_segments).float()
    cap = ops.K.lds_rows_max(dtype)
    for shape in a.shapes.split(","):
        r, c = (int(t) for t in shape.split("x"))
        if c > cap:
            continue
        plan, t_plan = wall(lambda: ops.BlockedPlan(g2.rowptr, g2.colind, g2.val, n, dtype, rows_per_block=r, lds_rows=c))
        y = ops.K.spmm_blocked(g2.rowptr, plan, xp, n, long_segments=g2.long_segments)
        err = float((y.float() - y_ref).norm() / y_ref.norm())
        report(f"row-block kernel {r} rows x {c} LDS slots", timed(lambda: ops.K.spmm_blocked(
            g2.rowptr, plan, xp, n, long_segments=g2.long_segments)), plan_ms=round(t_plan, 1),
            lds_fraction=round(plan.lds_fraction, 4), staged_rows_per_node=round(plan.staged_rows / n, 3),
            rel_diff_vs_plain=err)
        if a.ablate:
            for mask, what in ((1, "no staging loads"), (2, "no LDS entries"), (4, "no gathered entries"),
                               (6, "staging only"), (5, "LDS entries only, nothing staged"), (3, "gathered entries only")):
                os.environ["SGF_SPMM_BLK_DEBUG"] = str(mask)
                report(f"  ablation [{what}] {r}x{c}", timed(lambda: ops.K.spmm_blocked(
                    g2.rowptr, plan, xp, n, long_segments=g2.long_segments)))
            del os.environ["SGF_SPMM_BLK_DEBUG"]
        del plan


if __name__ == "__main__":
    main()
