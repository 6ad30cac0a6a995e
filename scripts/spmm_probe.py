#!/usr/bin/env python
"""SpMM-only measurements at ogbn-products scale (used for profiles/ and DESIGN.md, not by bench.py).

    python scripts/spmm_probe.py [--dtype bf16] [--windows 512,4096,32768] [--reps 5]

For the uniform random graph and for locality-structured graphs of the same size (90 % of the
undirected pairs within ~N(0, window) node ids), prints one JSON line per graph: launch time of
sgf_spmm (HIP events on the launch stream), algorithmic bytes, gather bytes (every
stored entry fetching a d-wide row) and the corresponding GB/s.  Run it under
`rocprofv3 --pmc ...` to attach TCC hit / miss and FETCH_SIZE counters to the same launches.
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sgformer_amd import ops, synth  # noqa: E402


def measure(ei, n, d, dtype, reps, dev, split=True):
    graph = ops.CSRGraph(ei, n, validate=False)
    nnz = int(ei.shape[1])
    segs = graph.long_segments if split else 0
    x = torch.randn(n, d, device=dev).to(dtype)
    for _ in range(2):
        ops.K.spmm(graph.rowptr, graph.colind, graph.val, x, n, long_segments=segs)
    evs = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        ops.K.spmm(graph.rowptr, graph.colind, graph.val, x, n, long_segments=segs)
        b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in evs)[len(evs) // 2]
    s = x.element_size()
    alg = nnz * 8 + (n + 1) * 8 + 2 * n * d * s
    gat = nnz * (8 + d * s) + (n + 1) * 8 + n * d * s
    lens = graph.rowptr[1:] - graph.rowptr[:-1]
    return {"nnz": nnz, "max_row": int(lens.max()), "long_segments": segs, "launch_ms": round(ms, 4), "algorithmic_GBps": round(alg / ms / 1e6, 1),
            "frac_of_8TBps": round(alg / ms / 1e6 / 8000.0, 4), "gather_GBps": round(gat / ms / 1e6, 1),
            "algorithmic_bytes": alg, "gather_bytes": gat}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16", choices=["f32", "bf16"])
    ap.add_argument("--windows", default="512,4096,32768")
    ap.add_argument("--locality", type=float, default=0.9)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--workload", default="ogbn-products")
    ap.add_argument("--skip-uniform", action="store_true")
    ap.add_argument("--hidden", type=int, default=0, help="override the feature width d")
    ap.add_argument("--skewed", default="", help="comma list of gamma values: power-law graphs, split vs unsplit")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    dtype = torch.float32 if args.dtype == "f32" else torch.bfloat16
    n, avg_deg, _, _, d = synth.SHAPES[args.workload]
    d = args.hidden or d
    if not args.skip_uniform:
        ei = synth.synthetic_graph(n, avg_deg, seed=123, device=dev)
        print(json.dumps({"graph": "uniform", "d": d, "dtype": args.dtype, **measure(ei, n, d, dtype, args.reps, dev)}), flush=True)
        del ei
    for gamma in [float(v) for v in args.skewed.split(",") if v]:
        ei = synth.synthetic_graph_skewed(n, avg_deg, gamma=gamma, seed=123, device=dev)
        for split in (False, True):
            print(json.dumps({"graph": f"skewed gamma={gamma}", "split_long_rows": split, "d": d, "dtype": args.dtype,
                              **measure(ei, n, d, dtype, args.reps, dev, split=split)}), flush=True)
        del ei
        ops.graph_cache.clear()
        torch.cuda.empty_cache()
    for w in [int(v) for v in args.windows.split(",") if v]:
        ei = synth.synthetic_graph_local(n, avg_deg, locality=args.locality, window=w, seed=123, device=dev)
        print(json.dumps({"graph": f"local p={args.locality} window={w}", "dtype": args.dtype,
                          **measure(ei, n, d, dtype, args.reps, dev)}), flush=True)
        del ei
        ops.graph_cache.clear()
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
