#!/usr/bin/env python
"""A/B of the wave-level SpMM kernels on ONE box (launch times vary +-5 % from box to box):
k_spmm_wave (r01: 64-bit addressing, wave per row), k_spmm_row (buffer addressing), k_spmm_seg /
k_spmm_seg_bf16x2 (flattened stream; two bf16 rows per load).  Interleaved, median of 9."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sgformer_amd import ops, synth  # noqa: E402

VARIANTS = {"k_spmm_wave (r01)": "wave", "k_spmm_row": "row", "k_spmm_seg": "seg", "k_spmm_seg_bf16x2": "seg2"}


def main():
    dev = torch.device("cuda:0")
    graph = sys.argv[1] if len(sys.argv) > 1 else "uniform"
    dtype = torch.bfloat16 if (len(sys.argv) < 3 or sys.argv[2] == "bf16") else torch.float32
    n, deg = 2449029, 50.5
    gen = synth.synthetic_graph_community if graph == "community" else synth.synthetic_graph
    ei = gen(n, deg, seed=123, device=dev)
    if graph == "community":
        _, inv, _ = ops.K.reorder(ei, n, *ops.REORDER_ITERS)
        ei = inv.long()[ei]
    g = ops.CSRGraph(ei, n, validate=False)
    del ei
    x = torch.randn(n, 256, device=dev).to(dtype)
    times = {k: [] for k in VARIANTS}
    for rep in range(10):
        for name, force in VARIANTS.items():
            os.environ["SGF_SPMM_KERNEL"] = force        # (seg2 falls back to row for fp32 storage)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            ops.K.spmm(g.rowptr, g.colind, g.val, x, n, long_segments=g.long_segments)
            b.record()
            torch.cuda.synchronize()
            if rep:
                times[name].append(a.elapsed_time(b))
    print(json.dumps({"graph": graph + (" (sgf_reorder order)" if graph == "community" else ""), "dtype": str(dtype),
                      "median_ms": {k: round(sorted(v)[len(v) // 2], 3) for k, v in times.items()}}))


if __name__ == "__main__":
    main()
