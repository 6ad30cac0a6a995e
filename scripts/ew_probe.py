#!/usr/bin/env python
"""Timings of the streaming elementwise / normalisation kernels at the ogbn-products shape (bf16, d = 256)."""
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
xs = [torch.randn(n, d, device=dev).bfloat16() for _ in range(7)]
x, r, g = xs[0], xs[1], xs[2]
gamma, beta = torch.rand(d, device=dev) + 0.5, torch.randn(d, device=dev)
mean, var, _ = ops.batch_stats(x)
rstd = torch.rsqrt(var + 1e-5)
K = ops.K
out = {}
def rec(name, fn, tensors):
    ms = timed(fn)
    out[name] = {"ms": round(ms, 4), "TBps": round(tensors * T / ms, 3)}
rec("sum_n (7)", lambda: K.sum_n(xs), 8)
rec("copy_", lambda: r.copy_(x), 2)
rec("colstats", lambda: K.colstats(x, mean), 1)
rec("bn_apply (+res, relu)", lambda: K.bn_apply(x, mean, rstd, gamma, beta, r, True), 3)
st = K.bn_bwd_stats(g, x, mean, rstd, gamma, beta, True)
rec("bn_bwd_stats", lambda: K.bn_bwd_stats(g, x, mean, rstd, gamma, beta, True), 2)
rec("bn_bwd_apply", lambda: K.bn_bwd_apply(g, x, mean, rstd, gamma, beta, True, st, 1.0 / n, True), 3)
y, mu, rs = K.ln_fwd(x, r, 0.5, 0.5, gamma, beta, True, 1e-5)
rec("ln_fwd (+res, relu)", lambda: K.ln_fwd(x, r, 0.5, 0.5, gamma, beta, True, 1e-5), 3)
rec("ln_fwd (stem)", lambda: K.ln_fwd(x, None, 1.0, 0.0, gamma, beta, True, 1e-5), 2)
rec("ln_bwd (+res)", lambda: K.ln_bwd(g, y, x, r, 0.5, 0.5, gamma, True, mu, rs), 5)
print(json.dumps(out))
