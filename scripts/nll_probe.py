#!/usr/bin/env python
"""Timing of the fused loss kernels (sgf_nll_fwd / _bwd) at the ogbn-products shape: 1.22 M training rows of 47 fp32 logits."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sgformer_amd import ops, synth  # noqa: E402

def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts) // 2]

dev = torch.device("cuda:0")
n, c = 2449029, 47
_, y, idx = synth.synthetic_task(n, 4, c, seed=123)
y, idx = y.to(dev), idx.to(dev)
logits = torch.randn(n, c, device=dev)
g = torch.ones(1, device=dev)
print(json.dumps({"nll_fwd_ms": timed(lambda: ops.K.nll_fwd(logits, y, idx)),
                  "nll_bwd_ms": timed(lambda: ops.K.nll_bwd(logits, y, idx, g, 1.0 / idx.numel()))}))
