#!/usr/bin/env python
"""Timing ablations of k_rowgemm_bf16 (SGF_ROWGEMM_DEBUG bits: 1 no MFMA, 2 no stores, 4 no re-loads)."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sgformer_amd import _lib, ops  # noqa: E402

def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts) // 2]

dev = torch.device("cuda:0")
n, d = 2449029, 256
a = torch.randn(n, d, device=dev).bfloat16()
w = (torch.randn(d, d, device=dev) / 16).bfloat16()
bias = torch.randn(d, device=dev)
out = {}
for flags in (0, 2, 4, 6):
    os.environ["SGF_ROWGEMM_DEBUG"] = str(flags); _lib.load().sgf_reload_env()
    out[f"dbg={flags}"] = round(timed(lambda: ops.K.gcn_epilogue_stats(a, w, bias)), 4)
os.environ["SGF_ROWGEMM_DEBUG"] = "0"; _lib.load().sgf_reload_env()
y = torch.empty_like(a)
out["copy_ (torch)"] = round(timed(lambda: y.copy_(a)), 4)
print(json.dumps(out))
