#!/usr/bin/env python
"""Timing ablations of sgf_gcn_bn_bwd_dx at the ogbn-products shape (SGF_GCN_BWD_DEBUG bits: 1 no element-wise work,
2 no matrix-core work, 4 no stores, 8 no re-loads), with and without the per-tile rendezvous."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sgformer_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")
n, d = 2449029, 256
g = torch.Generator(device=dev).manual_seed(1)
K = ops.K
gy = torch.randn(n, d, device=dev, generator=g).bfloat16()
z = torch.randn(n, d, device=dev, generator=g).bfloat16()
w = (torch.randn(d, 2 * d, device=dev, generator=g) / (2 * d) ** 0.5).bfloat16()
mean = torch.randn(d, device=dev, generator=g) * 0.2
rstd = 1.0 / (1.0 + torch.rand(d, device=dev, generator=g))
gamma = 1.0 + 0.3 * torch.randn(d, device=dev, generator=g)
beta = 0.2 * torch.randn(d, device=dev, generator=g)
stats = K.bn_bwd_stats(gy, z, mean, rstd, gamma, beta, True)
acc0 = K.gcn_bn_bwd_dx(gy, z, mean, rstd, gamma, beta, True, stats, 1.0 / n, True, w, None, last=False, add_gy=True)[2]


def run():
    return K.gcn_bn_bwd_dx(gy, z, mean, rstd, gamma, beta, True, stats, 1.0 / n, True, w, acc0, last=False, add_gy=True)


def timed(reps=8):
    run(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); run(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return round(sorted(ts)[len(ts) // 2], 3)


out = {}
for sync in (1, 0):
    for dbg in (0, 1, 2, 3, 4, 8, 12, 15):
        os.environ["SGF_GCN_BWD_DEBUG"] = str(dbg)
        os.environ["SGF_GCN_BWD_SYNC"] = str(sync)
        _lib.load().sgf_reload_env()
        out[f"sync={sync} dbg={dbg}"] = timed()
print(json.dumps(out, indent=1))
