#!/usr/bin/env python
"""Tile SpMM: what the XCD's L2 keeps.  Times (and, under `rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum` / `--pmc FETCH_SIZE`,
counts) sgf_spmm_tile under cache-policy experiments of the PROBES library and XCD chunk sizes.

    make PROBES=1 BUILD=build_probes LIB=sgformer_amd/lib/libsgf_probes.so
    python scripts/tile_l2_probe.py                      # medians
    python scripts/tile_l2_probe.py --once               # ONE launch per setting, in the printed order (for counter passes)
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sgformer_amd import _lib  # noqa: E402

_lib.LIB_PATH = os.path.join(ROOT, "sgformer_amd", "lib", "libsgf_probes.so")
from sgformer_amd import ops, synth  # noqa: E402

SETTINGS = [  # (debug mask, chunk, what[, column halves])
    (0, 64, "as shipped"),
    (0, 64, "two column halves", True),
    (0, 104, "two column halves, chunk 104", True),
    (0, 208, "two column halves, chunk 208", True),
    (4096 + 8 + 512, 64, "two column halves, far gathers nt, y stores nt, packed tiles nt", True),
    (4096 + 8 + 512, 104, "two column halves, far gathers nt, y stores nt, packed tiles nt, chunk 104", True),
    (4096 + 8 + 512, 208, "two column halves, far gathers nt, y stores nt, packed tiles nt, chunk 208", True),
    (4096 + 8 + 512, 416, "two column halves, far gathers nt, y stores nt, packed tiles nt, chunk 416", True),
]


def apply(dbg, chunk):
    os.environ["SGF_SPMM_TILE_DEBUG"] = str(dbg)
    os.environ["SGF_SPMM_TILE_CHUNK"] = str(chunk)
    _lib.load().sgf_reload_env()


def timed(fn, reps=11, warm=6):
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
    ap.add_argument("--graph", default="community", choices=["community", "powerlaw"])
    ap.add_argument("--once", action="store_true")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    n = 2449029
    gen = {"community": synth.synthetic_graph_community, "powerlaw": synth.synthetic_graph_community_powerlaw}[a.graph]
    ei = gen(n, 50.5, seed=123, device=dev)
    x = torch.randn(n, 256, device=dev).to(torch.bfloat16)
    perm, inv, comm = ops.K.reorder(ei, n, *ops.REORDER_ITERS)
    g2 = ops.CSRGraph(inv.long()[ei], n, validate=False)
    del ei
    blk = ops.K.tile_blocks(comm[perm.long()].contiguous(), n, 128, dev)
    plan = ops.TilePlan(g2.rowptr, g2.colind, g2.val, n, blk, cap=512, min_count=2)
    y = torch.empty_like(x)
    full = lambda: ops.K.spmm_tile(plan, x, n, out=y)  # noqa: E731

    def halves():
        ops.K.spmm_tile(plan, x[:, :128], n, out=y[:, :128])
        ops.K.spmm_tile(plan, x[:, 128:], n, out=y[:, 128:])

    rows = []
    for rnd in range(1 if a.once else 2):
        for i, st in enumerate(SETTINGS):
            dbg, chunk, what = st[:3]
            run = halves if len(st) > 3 and st[3] else full
            apply(dbg, chunk)
            if a.once:
                run()
                torch.cuda.synchronize()
                rows.append({"launch": i, "dbg": dbg, "chunk": chunk, "what": what})
            else:
                rows.append({"round": rnd, "dbg": dbg, "chunk": chunk, "what": what, "ms": round(timed(run), 4)})
            print(json.dumps(rows[-1]), flush=True)
    if a.out:
        with open(a.out, "a") as f:
            for r in rows:
                f.write(json.dumps({"graph": a.graph, **r}) + "\n")


if __name__ == "__main__":
    main()
