#!/usr/bin/env python
"""Timing of the fused head kernels (sgf_combine_fc_fwd / _bwd) at the ogbn-products shape."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sgformer_amd import ops  # noqa: E402

def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts) // 2]

dev = torch.device("cuda:0")
n, d, c = 2449029, 256, 47
x1 = torch.randn(n, d, device=dev).bfloat16()
x2 = torch.randn(n, d, device=dev).bfloat16()
w = torch.randn(c, d, device=dev) / 16
b = torch.randn(c, device=dev)
g = torch.randn(n, c, device=dev)
out = {"fwd_ms": timed(lambda: ops.K.combine_fc_fwd(x1, 0.5, x2, 0.5, w, b)),
       "bwd_ms": timed(lambda: ops.K.combine_fc_bwd(g, w, 0.5, 0.5, torch.bfloat16))}
print(json.dumps(out))
