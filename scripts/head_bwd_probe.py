#!/usr/bin/env python
"""sgf_combine_fc_bwd_g at ogbn-products size (N = 2.45 M, d = 256, 47 classes, bf16): median launch time, with and without
a row map, and the result against fp64 of the same inputs."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sgformer_amd import _lib  # noqa: E402

if os.environ.get("SGF_PROBE_LIB"):
    _lib.LIB_PATH = os.path.join(ROOT, os.environ["SGF_PROBE_LIB"])
from sgformer_amd import ops  # noqa: E402

K = ops.K


def timed(fn, reps=15, warm=6):
    ts = []
    for i in range(reps + warm):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record()
        torch.cuda.synchronize()
        if i >= warm:
            ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2]


def main():
    dev = torch.device("cuda:0")
    n, d, c = 2449029, 256, 47
    g = torch.Generator(device=dev).manual_seed(0)
    dl = torch.randn(n, c, device=dev, generator=g) * 1e-3
    w = torch.randn(c, d, device=dev, generator=g) * 0.1
    rmap = torch.randperm(n, device=dev, generator=g).int()
    out = {}
    for name, rm in (("no row map", None), ("row map", rmap)):
        fn = lambda: K.combine_fc_bwd_g(dl, w, 0.8, 0.2, rm)  # noqa: E731
        ms = timed(fn)
        dx1, dx2, gp = fn()
        sel = torch.arange(0, n, 997, device=dev)
        src = dl[rm.long()[sel]] if rm is not None else dl[sel]
        ref = src.to(torch.bfloat16).double() @ w.to(torch.bfloat16).double()
        e1 = float((dx1[sel].double() - 0.8 * ref).abs().max() / (0.8 * ref).abs().max())
        e2 = float((dx2[sel].double() - 0.2 * ref).abs().max() / (0.2 * ref).abs().max())
        gb = (n * c * 4 + 2 * n * d * 2 + n * 48 * 2) / 1e9
        out[name] = {"ms": round(ms, 4), "TB/s": round(gb / ms, 3), "rel_err_dx1": e1, "rel_err_dx2": e2}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
