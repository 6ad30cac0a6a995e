#!/usr/bin/env python
"""sgf_gcn_epilogue_stats / _dx against torch.nn.functional.linear (hipBLASLt) + sgf_colstats at the
ogbn-products shape: correctness vs an fp64 host product on a slice, then interleaved timings."""
import json
import sys
import os

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sgformer_amd import ops  # noqa: E402


def timed(fn, reps=10):
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
    return sorted(ts)[len(ts) // 2]


def main():
    dev = torch.device("cuda:0")
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2449029
    d = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    g = torch.Generator(device=dev).manual_seed(1)
    a = torch.randn(n, d, device=dev, generator=g).bfloat16()
    w = (torch.randn(d, d, device=dev, generator=g) / d ** 0.5).bfloat16()
    bias = torch.randn(d, device=dev, generator=g)
    shift = torch.randn(d, device=dev, generator=g) * 0.1
    K = ops.K
    y, st = K.gcn_epilogue_stats(a, w, bias, shift, want_stats=True)
    dx = K.gcn_epilogue_dx(a, w)
    torch.cuda.synchronize()
    m = min(n, 4096)
    sl = slice(n - m, n)
    ref = a[sl].double() @ w.double().t() + bias.double()
    err_y = (y[sl].double() - ref).abs().max().item()
    refdx = a[sl].double() @ w.double()
    err_dx = (dx[sl].double() - refdx).abs().max().item()
    yd = y.double() - shift.double()
    st_ref = torch.cat([yd.sum(0), (yd * yd).sum(0)])
    err_st = ((st.double() - st_ref).abs() / st_ref.abs().clamp_min(1.0)).max().item()
    y_lib = torch.nn.functional.linear(a, w, bias.bfloat16())
    mism = (y_lib != y).float().mean().item()
    out = {"n": n, "d": d, "max_abs_err_y_vs_fp64": err_y, "max_abs_err_dx_vs_fp64": err_dx,
           "max_rel_err_stats": err_st, "fraction_of_y_differing_from_hipblaslt": mism}
    bb = bias.bfloat16()
    out["ms"] = {
        "F.linear (hipBLASLt)": timed(lambda: torch.nn.functional.linear(a, w, bb)),
        "sgf_gcn_epilogue_stats, no stats": timed(lambda: K.gcn_epilogue_stats(a, w, bias)),
        "sgf_gcn_epilogue_stats, stats": timed(lambda: K.gcn_epilogue_stats(a, w, bias, shift, want_stats=True)),
        "sgf_colstats": timed(lambda: K.colstats(y, shift)),
        "a @ w (hipBLASLt dX)": timed(lambda: a @ w),
        "sgf_gcn_epilogue_dx": timed(lambda: K.gcn_epilogue_dx(a, w)),
    }
    gb = 2 * n * d * 2 / 1e9
    out["TBps"] = {k: round(gb / v, 3) for k, v in out["ms"].items() if "colstats" not in k}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
