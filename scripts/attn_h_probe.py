#!/usr/bin/env python
"""Timings of the attention-from-input passes (sgf_attn_h_*) and sgf_gram at the ogbn-products shape (bf16, d = 256)."""
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
n, d = 2449029, 256
T = n * d * 2 / 1e9
h = torch.randn(n, d, device=dev).bfloat16()
g = torch.randn(n, d, device=dev).bfloat16()
M = torch.randn(d, d, device=dev) / 16
D = torch.randn(d, d, device=dev) / 16
m, w, ds = torch.randn(d, device=dev), torch.rand(d, device=dev) / d, torch.randn(d, device=dev)
beta = torch.full((1,), 4.0, device=dev)
K = ops.K
out_t, den = K.attn_h_fwd(h, M, m, w, beta)
res = {}
def rec(name, fn, tensors):
    ms = timed(fn)
    res[name] = {"ms": round(ms, 4), "TBps": round(tensors * T / ms, 3)}
rec("attn_h_fwd", lambda: K.attn_h_fwd(h, M, m, w, beta), 2)
rec("attn_h_bwd_reduce", lambda: K.attn_h_bwd_reduce(h, g, out_t, den), 3)
rec("attn_h_bwd_apply (2 passes)", lambda: K.attn_h_bwd_apply(h, g, out_t, den, M, w, D, ds), 6)
rec("attn_h_bwd_pre (B1 + row scalars)", lambda: K.attn_h_bwd_pre(g, out_t, den, M, w), 3)
rs = K.attn_h_bwd_pre(g, out_t, den, M, w)
rec("attn_h_bwd_reduce_scaled", lambda: K.attn_h_bwd_reduce_scaled(h, g, rs), 2)
rec("attn_h_bwd_post (B2)", lambda: K.attn_h_bwd_post(h, D, ds), 3)
rec("gram", lambda: K.gram(g, h), 2)
print(json.dumps(res))
