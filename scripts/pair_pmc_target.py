#!/usr/bin/env python
"""PMC target: ONLY the paired launches of round 4 at the ogbn-products shape — sgf_gcn_epilogue_dx2 (k_rowgemm_bf16) and
sgf_gram2 (k_reduce_bf16<256, 2, 16>) — three launches each, so that scripts/pmc_kernel_summary.py averages nothing else.
    bash scripts/pmc_passes.sh gpurun_out/r4_pair_pmc python scripts/pair_pmc_target.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sgformer_amd import ops  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    n, d = 2449029, 256
    g = torch.Generator(device=dev).manual_seed(1)
    dz = torch.randn(n, d, device=dev, generator=g).bfloat16()
    y = torch.randn(n, d, device=dev, generator=g).bfloat16()
    x0 = torch.randn(n, d, device=dev, generator=g).bfloat16()
    w = (torch.randn(d, 2 * d, device=dev, generator=g) / (2 * d) ** 0.5).bfloat16()
    dw = torch.empty(d, 2 * d, device=dev)
    for _ in range(3):
        ops.K.gcn_epilogue_dx2(dz, w[:, :d], w[:, d:], True)
    for _ in range(3):
        ops.K.gram2(dz, y, x0, dw[:, :d], dw[:, d:], want_colsum=True)
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
