#!/usr/bin/env python
"""Induced-subgraph step of large/main-batch.py at ogbn-products scale: sgformer_amd.batching.subgraph
(GPU, sgf_subgraph_*) against the host-side torch implementation of the PyG semantics the reference
runs per batch.   python scripts/subgraph_probe.py [--batch 100000]"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sgformer_amd import batching, synth  # noqa: E402


def host_subgraph(subset, ei, n):
    mask_n = torch.zeros(n, dtype=torch.bool)
    mask_n[subset] = True
    keep = mask_n[ei[0]] & mask_n[ei[1]]
    idx = torch.zeros(n, dtype=torch.int64)
    idx[subset] = torch.arange(subset.numel())
    return idx[ei[:, keep]]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=100000)   # large/run.sh:19 (amazon2m)
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    n, avg_deg, _, _, _ = synth.SHAPES["ogbn-products"]
    dev = torch.device("cuda:0")
    ei = synth.synthetic_graph(n, avg_deg, seed=123, device=dev).cpu()
    nnz = int(ei.shape[1])
    g = torch.Generator().manual_seed(0)
    perm = torch.randperm(n, generator=g)
    subsets = [perm[i * args.batch:(i + 1) * args.batch] for i in range(args.reps + 1)]
    batching.subgraph(subsets[0], ei, num_nodes=n, relabel_nodes=True)      # stages edge_index, warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    outs = [batching.subgraph(s, ei, num_nodes=n, relabel_nodes=True)[0] for s in subsets[1:]]
    torch.cuda.synchronize()
    gpu_ms = (time.perf_counter() - t0) / args.reps * 1e3
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    t0 = time.perf_counter()
    ref = host_subgraph(subsets[1], ei, n)
    cpu_ms = (time.perf_counter() - t0) * 1e3
    assert torch.equal(outs[0].cpu(), ref)
    print(json.dumps({"edges": nnz, "batch_nodes": args.batch, "kept_edges": int(ref.shape[1]),
                      "gpu_ms_per_batch": round(gpu_ms, 3), "edges_per_s_gpu": round(nnz / gpu_ms * 1e3),
                      "algorithmic_GBps": round(2 * nnz * 16 / gpu_ms / 1e6, 1),
                      "host_ms_per_batch": round(cpu_ms, 1), "speedup": round(cpu_ms / gpu_ms, 1),
                      "bit_exact": True}))


if __name__ == "__main__":
    main()
