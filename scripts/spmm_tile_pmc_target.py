#!/usr/bin/env python
"""Profiling target: a few launches of sgf_spmm_tile on the re-ordered community graph, nothing else on the GPU clock
(counters in their own passes, never combined with tracing — scripts/pmc_passes.sh)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sgformer_amd import ops, synth  # noqa: E402

dev = torch.device("cuda:0")
n, deg, d = 2449029, 50.5, 256
ei = synth.synthetic_graph_community(n, deg, seed=123, device=dev)
perm, inv, comm = ops.K.reorder(ei, n, *ops.REORDER_ITERS)
g = ops.CSRGraph(inv.long()[ei], n, validate=False)
del ei
x = torch.randn(n, d, device=dev).to(torch.bfloat16)
g.blk_row = ops.K.tile_blocks(comm[perm.long()].contiguous(), n, 128, dev)
plan = ops.TilePlan(g.rowptr, g.colind, g.val, n, g.blk_row, cap=512, min_count=int(os.environ.get("MIN_COUNT", "2")))
for _ in range(int(os.environ.get("REPS", "3"))):
    ops.K.spmm_tile(plan, x, n)
torch.cuda.synchronize()
