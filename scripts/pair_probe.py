#!/usr/bin/env python
"""sgf_gcn_epilogue_dx2: both input gradients of the two-operand Linear from one HBM read of dy.

A PAIRED launch (blocks b and b + 8 of one XCD walk the same row tiles with different resident weight blocks) against
two separate launches, at the ogbn-products shape: bit-equality of the results, interleaved timings."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sgformer_amd import ops  # noqa: E402


def timed(fn, reps=12):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2], ts[0], ts[-1]


def main():
    dev = torch.device("cuda:0")
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2449029
    d = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    g = torch.Generator(device=dev).manual_seed(1)
    dy = torch.randn(n, d, device=dev, generator=g).bfloat16()
    w = (torch.randn(d, 2 * d, device=dev, generator=g) / d ** 0.5).bfloat16()
    w1, w2 = w[:, :d], w[:, d:]
    K = ops.K
    a1, a2 = K.gcn_epilogue_dx2(dy, w1, w2, pair=True)
    b1, b2 = K.gcn_epilogue_dx2(dy, w1, w2, pair=False)
    torch.cuda.synchronize()
    out = {"n": n, "d": d, "paired_equals_separate": bool(torch.equal(a1, b1) and torch.equal(a2, b2))}
    m = min(n, 4096)
    ref = dy[-m:].double() @ w2.double()
    out["max_abs_err_dx2_vs_fp64"] = (a2[-m:].double() - ref).abs().max().item()
    del a1, a2, b1, b2
    ms = {}
    for rep in range(2):
        ms[f"two launches #{rep}"] = timed(lambda: K.gcn_epilogue_dx2(dy, w1, w2, pair=False))
        ms[f"paired launch #{rep}"] = timed(lambda: K.gcn_epilogue_dx2(dy, w1, w2, pair=True))
        ms[f"one dx launch #{rep}"] = timed(lambda: K.gcn_epilogue_dx(dy, w1))
    out["ms (median, min, max)"] = ms
    t = n * d * 2 / 1e9
    out["TBps on 3T (read once, write twice)"] = {k: round(3 * t / v[0], 3) for k, v in ms.items() if "one" not in k}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
