#!/usr/bin/env python
"""k_spmm_seg_bf16x2 on the re-ordered community graph for several XCD chunk sizes (SGF_SPMM_CHUNK_ROWS)."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sgformer_amd import ops, synth  # noqa: E402
dev = torch.device("cuda:0")
n, deg = 2449029, 50.5
ei = synth.synthetic_graph_community(n, deg, seed=123, device=dev)
_, inv, _ = ops.K.reorder(ei, n, *ops.REORDER_ITERS)
ei = inv.long()[ei]
g = ops.CSRGraph(ei, n, validate=False)
del ei
x = torch.randn(n, 256, device=dev).bfloat16()
os.environ["SGF_SPMM_KERNEL"] = "seg2"
out = {}
for rows in (128, 512, 1024, 2048, 4096, 8192, 16384, 65536):
    os.environ["SGF_SPMM_CHUNK_ROWS"] = str(rows)
    ts = []
    for rep in range(8):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        ops.K.spmm(g.rowptr, g.colind, g.val, x, n, long_segments=g.long_segments)
        b.record()
        torch.cuda.synchronize()
        if rep:
            ts.append(a.elapsed_time(b))
    out[rows] = round(sorted(ts)[len(ts) // 2], 3)
print(json.dumps(out))
