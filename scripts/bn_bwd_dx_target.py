#!/usr/bin/env python
"""A few launches of sgf_gcn_bn_bwd_dx (and, for comparison, of sgf_gcn_epilogue_dx and sgf_gcn_epilogue_cat) at the
ogbn-products shape: the target of scripts/pmc_passes.sh for the counters of the row-streaming kernels."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sgformer_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2449029
d = int(sys.argv[2]) if len(sys.argv) > 2 else 256
g = torch.Generator(device=dev).manual_seed(1)
K = ops.K
gy = torch.randn(n, d, device=dev, generator=g).bfloat16()
z = torch.randn(n, d, device=dev, generator=g).bfloat16()
w = (torch.randn(d, 2 * d, device=dev, generator=g) / (2 * d) ** 0.5).bfloat16()
mean = torch.randn(d, device=dev, generator=g) * 0.2
rstd = 1.0 / (1.0 + torch.rand(d, device=dev, generator=g))
gamma = 1.0 + 0.3 * torch.randn(d, device=dev, generator=g)
beta = 0.2 * torch.randn(d, device=dev, generator=g)
bias = torch.randn(d, device=dev, generator=g)
stats = K.bn_bwd_stats(gy, z, mean, rstd, gamma, beta, True)
acc = None
for i in range(4):
    _, _, acc = K.gcn_bn_bwd_dx(gy, z, mean, rstd, gamma, beta, True, stats, 1.0 / n, True, w, acc, last=False, add_gy=True)
for i in range(4):
    K.gcn_epilogue_dx(gy, w[:, :d])
for i in range(4):
    K.gcn_epilogue_cat(gy, z, w, bias, None, want_stats=False)
torch.cuda.synchronize()
