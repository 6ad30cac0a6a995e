#!/usr/bin/env python
"""Does the tile SpMM's gather working set fit the L2 when the features are processed in column slices?

The community generator's super-community (64 communities, ~10 k nodes) is 5.2 MB of bf16 rows at d = 256 — more than an
XCD's 4 MB of L2 (profiles/r05_spmm_pmc.md: 72 % of those gathers miss).  At d = 128 it is 2.6 MB.  This probe times
sgf_spmm_tile on [n, 256] in one launch against two launches over the column halves (views with ld = 256), same plan.

    python scripts/tile_slice_probe.py [--graph community|powerlaw] [--reps 10]
    rocprofv3 --pmc FETCH_SIZE -d ... -- python scripts/tile_slice_probe.py --reps 2      # traffic of both forms
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sgformer_amd import ops, synth  # noqa: E402


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record()
    for i in range(reps):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(reps))
    return ts[len(ts) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--graph", default="community", choices=["community", "powerlaw"])
    ap.add_argument("--n", type=int, default=2449029)
    ap.add_argument("--deg", type=float, default=50.5)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--tile", default="512,2,128")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    gen = {"community": synth.synthetic_graph_community, "powerlaw": synth.synthetic_graph_community_powerlaw}[a.graph]
    ei = gen(a.n, a.deg, seed=123, device=dev)
    n = a.n
    x = torch.randn(n, 256, device=dev).to(torch.bfloat16)
    perm, inv, comm = ops.K.reorder(ei, n, *ops.REORDER_ITERS)
    g2 = ops.CSRGraph(inv.long()[ei], n, validate=False)
    del ei
    xp = ops.gather_rows(x, perm)
    cap, mc, mr = (int(t) for t in a.tile.split(","))
    blk = ops.K.tile_blocks(comm[perm.long()].contiguous(), n, mr, dev)
    plan = ops.TilePlan(g2.rowptr, g2.colind, g2.val, n, blk, cap=cap, min_count=mc)
    y_full = torch.empty_like(xp)
    y_half = torch.empty_like(xp)

    def full():
        ops.K.spmm_tile(plan, xp, n, out=y_full)

    def halves():
        ops.K.spmm_tile(plan, xp[:, :128], n, out=y_half[:, :128])
        ops.K.spmm_tile(plan, xp[:, 128:], n, out=y_half[:, 128:])

    def half0():
        ops.K.spmm_tile(plan, xp[:, :128], n, out=y_half[:, :128])

    res = {"graph": a.graph, "tile_fraction": plan.tile_fraction, "nb": plan.nb,
           "full_ms": timed(full, a.reps), "two_halves_ms": timed(halves, a.reps), "half0_ms": timed(half0, a.reps)}
    full()
    halves()
    torch.cuda.synchronize()
    res["identical"] = bool(torch.equal(y_full, y_half))
    res["max_abs_diff"] = float((y_full.float() - y_half.float()).abs().max())
    print(json.dumps(res))
    if a.out:
        with open(a.out, "a") as f:
            f.write(json.dumps(res) + "\n")


if __name__ == "__main__":
    main()
