#!/usr/bin/env python
"""SpMM at hidden width 128 (papers100M recipe), bf16: k_spmm_sub (half-wave per row, 8 B per lane) against the
wave-per-row and stream kernels of the d = 256 path run with half their lanes idle (SGF_SPMM_KERNEL)."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sgformer_amd import ops, synth  # noqa: E402
dev = torch.device("cuda:0")
n, deg, d = 6000000, 29.0, 128
ei = synth.synthetic_graph(n, deg, seed=7, device=dev)
g = ops.CSRGraph(ei, n, validate=False)
nnz = int(ei.shape[1])
del ei
x = torch.randn(n, d, device=dev).bfloat16()
out = {"n": n, "nnz": nnz, "d": d}
ref = None
for name in ("sub", "", "row", "seg2", "sub", ""):
    os.environ["SGF_SPMM_KERNEL"] = name      # "" = the library's own choice
    ts = []
    for rep in range(7):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        y = ops.K.spmm(g.rowptr, g.colind, g.val, x, n, long_segments=g.long_segments)
        b.record()
        torch.cuda.synchronize()
        if rep:
            ts.append(a.elapsed_time(b))
    if ref is None:
        ref = y.float()
    ms = sorted(ts)[len(ts) // 2]
    out.setdefault(name or "default", []).append({"ms": round(ms, 3), "gather_TBps": round(nnz * d * 2 / ms / 1e9, 2),
                                     "relerr": float((y.float() - ref).norm() / ref.norm())})
print(json.dumps(out))
