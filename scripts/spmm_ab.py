#!/usr/bin/env python
"""A/B of the wave-level SpMM kernels on ONE box (launch times vary +-5 % from box to box):
k_spmm_wave (r01: 64-bit addressing, wave per row), k_spmm_row (buffer addressing), k_spmm_seg /
k_spmm_seg_bf16x2 (flattened stream; two bf16 rows per load), and — on the re-ordered graphs, bf16 — the matrix-core
row-block kernel k_spmm_tile_bf16.  Interleaved, median of 9.

    python scripts/spmm_ab.py [uniform|community|powerlaw] [bf16|f32]
(powerlaw: synth.synthetic_graph_community_powerlaw — skewed communities, local and global hubs, shuffled ids)"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sgformer_amd import ops, synth  # noqa: E402

VARIANTS = {"k_spmm_wave (r01)": "wave", "k_spmm_row": "row", "k_spmm_seg": "seg", "k_spmm_seg_bf16x2": "seg2"}


def main():
    dev = torch.device("cuda:0")
    graph = sys.argv[1] if len(sys.argv) > 1 else "uniform"
    dtype = torch.bfloat16 if (len(sys.argv) < 3 or sys.argv[2] == "bf16") else torch.float32
    n, deg = 2449029, 50.5
    gen = {"community": synth.synthetic_graph_community, "powerlaw": synth.synthetic_graph_community_powerlaw,
           "uniform": synth.synthetic_graph}[graph]
    ei = gen(n, deg, seed=123, device=dev)
    structured = graph != "uniform"
    plan = None
    if structured:
        perm, inv, comm = ops.K.reorder(ei, n, *ops.REORDER_ITERS)
        ei = inv.long()[ei]
    g = ops.CSRGraph(ei, n, validate=False)
    del ei
    x = torch.randn(n, 256, device=dev).to(dtype)
    variants = dict(VARIANTS)
    if structured and dtype == torch.bfloat16:
        g.blk_row = ops.K.tile_blocks(comm[perm.long()].contiguous(), n, ops.TILE_MAX_ROWS, dev)
        plan = ops.TilePlan(g.rowptr, g.colind, g.val, n, g.blk_row)
        if plan.tile_density < ops.TILE_SPARSE_DENSITY:          # the policy of ops.GraphView
            plan = ops.TilePlan(g.rowptr, g.colind, g.val, n, g.blk_row, min_count=ops.TILE_MIN_COUNT + 1)
        variants["k_spmm_tile_bf16"] = None
    times = {k: [] for k in variants}
    for rep in range(10):
        for name, force in variants.items():
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            if force is None:
                ops.K.spmm_tile(plan, x, n)
            else:
                os.environ["SGF_SPMM_KERNEL"] = force    # (seg2 falls back to row for fp32 storage)
                ops.K.spmm(g.rowptr, g.colind, g.val, x, n, long_segments=g.long_segments)
            b.record()
            torch.cuda.synchronize()
            if rep:
                times[name].append(a.elapsed_time(b))
    print(json.dumps({"graph": graph + (" (sgf_reorder order)" if structured else ""), "dtype": str(dtype),
                      "median_ms": {k: round(sorted(v)[len(v) // 2], 3) for k, v in times.items()}}))


if __name__ == "__main__":
    main()
