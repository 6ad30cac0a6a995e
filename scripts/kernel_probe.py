#!/usr/bin/env python
"""Stand-alone timing of every streaming kernel of libsgf at one shape (default: ogbn-products scale).

    python scripts/kernel_probe.py [--nodes 2449029] [--hidden 256] [--dtype bf16] [--reps 7] [--only apply,gram]

One JSON line per kernel: median launch time (HIP events on the launch stream), the kernel's ALGORITHMIC
bytes (the [n, d] tensors it has to read + write) and the resulting GB/s against the 8 TB/s
spec figure and the 6.3 TB/s a device copy reaches.  This is the loop to iterate a kernel in: the same
table from a whole training step is profiles/rNN_*_kernel_roofline.md (scripts/kernel_roofline.py).
The hipBLASLt GEMM of the same shape and a plain device copy are timed as yardsticks.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sgformer_amd import ops  # noqa: E402

SPEC_GBPS, COPY_GBPS = 8000.0, 6300.0


def _median_ms(fn, reps: int, dev: torch.device) -> float:
    fn()
    if dev.type != "cuda":                      # the CPU kernel table of the test-suite: wall clock
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            ts.append((time.perf_counter() - t0) * 1e3)
        return sorted(ts)[len(ts) // 2]
    fn()
    evs = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        evs.append((a, b))
    torch.cuda.synchronize(dev)
    return sorted(a.elapsed_time(b) for a, b in evs)[len(evs) // 2]


def cases(n: int, d: int, dtype: torch.dtype, dev: torch.device):
    """(name, tensors read+written in units of [n, d] activations, thunk).  Inputs are allocated once."""
    K = ops.K
    g = torch.Generator(device="cpu").manual_seed(1)

    def act():
        return torch.randn(n, d, generator=g).to(dev, dtype)

    def vec(scale=1.0, shift=0.0):
        return (torch.randn(d, generator=g) * scale + shift).to(dev)

    h, x2, gy, o = act(), act(), act(), act()
    den = (torch.rand(n, 1, generator=g) + 1.0).to(dev)
    M = (torch.randn(d, d, generator=g) / d ** 0.5).to(dev)
    D = (torch.randn(d, d, generator=g) / d ** 0.5).to(dev)
    m_, w_, ds_ = vec(), vec(0.01, 1.0 / d), vec()
    beta = torch.ones(1, device=dev)
    gamma, bbeta = vec(0.1, 1.0), vec(0.1)
    mean, rstd = vec(0.1), vec(0.05, 1.0).abs()
    shift = torch.zeros(d, device=dev)
    y_ln, mu_ln, rs_ln = K.ln_fwd(h, x2, 0.5, 0.5, gamma, bbeta, True, 1e-5)
    bstats = K.bn_bwd_stats(gy, h, mean, rstd, gamma, bbeta, True)
    wlin = (torch.randn(d, d, generator=g) / d ** 0.5).to(dev, dtype)
    blin = vec().to(dtype)
    operands = [h, x2, gy, o, y_ln, h, x2]

    out = [
        ("copy (yardstick)", 2, lambda: h.clone()),
        ("hipBLASLt [n,d]x[d,d]+bias (yardstick)", 2, lambda: torch.addmm(blin, h, wlin.t())),
        ("sgf_gram(h, h)", 1, lambda: K.gram(h, h)),
        ("sgf_gram(gy, h)  (dW, db)", 2, lambda: K.gram(gy, h)),
        ("sgf_attn_h_fwd", 2, lambda: K.attn_h_fwd(h, M, m_, w_, beta)),
        ("sgf_attn_h_bwd_reduce", 3, lambda: K.attn_h_bwd_reduce(h, gy, o, den)),
        ("sgf_attn_h_bwd_apply", 4, lambda: K.attn_h_bwd_apply(h, gy, o, den, M, w_, D, ds_)),
        ("sgf_ln_fwd (residual, relu)", 3, lambda: K.ln_fwd(h, x2, 0.5, 0.5, gamma, bbeta, True, 1e-5)),
        ("sgf_ln_bwd (residual, relu)", 6, lambda: K.ln_bwd(gy, y_ln, h, x2, 0.5, 0.5, gamma, True, mu_ln, rs_ln)),
        ("sgf_colstats", 1, lambda: K.colstats(h, shift)),
        ("sgf_bn_apply (residual, relu)", 3, lambda: K.bn_apply(h, mean, rstd, gamma, bbeta, x2, True)),
        ("sgf_bn_bwd_stats", 2, lambda: K.bn_bwd_stats(gy, h, mean, rstd, gamma, bbeta, True)),
        ("sgf_bn_bwd_apply", 3, lambda: K.bn_bwd_apply(gy, h, mean, rstd, gamma, bbeta, True, bstats, 1.0 / n, True)),
        ("sgf_axpby", 3, lambda: K.axpby(h, 0.5, x2, 0.5)),
        ("sgf_sum_n (7 operands)", 8, lambda: K.sum_n(operands)),
        ("sgf_dropout (+residual)", 3, lambda: K.dropout(h, x2, 0.5, 1234)),
        ("sgf_colsum", 1, lambda: K.colsum(h)),
    ]
    return out


def run(n: int, d: int, dtype: torch.dtype, dev, reps: int = 7, only=()):
    dev = torch.device(dev)
    unit = n * d * torch.empty((), dtype=dtype).element_size()
    rows = []
    for name, tensors, fn in cases(n, d, dtype, dev):
        if only and not any(s in name for s in only):
            continue
        ms = _median_ms(fn, reps, dev)
        gbps = tensors * unit / ms / 1e6
        rows.append({"kernel": name, "ms": round(ms, 4), "algorithmic_GB": round(tensors * unit / 1e9, 3),
                     "GBps": round(gbps, 1), "frac_of_8TBps": round(gbps / SPEC_GBPS, 3),
                     "frac_of_copy_ceiling": round(gbps / COPY_GBPS, 3)})
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nodes", type=int, default=2449029)
    ap.add_argument("--hidden", type=int, default=256)
    ap.add_argument("--dtype", default="bf16", choices=["f32", "bf16"])
    ap.add_argument("--reps", type=int, default=7)
    ap.add_argument("--only", default="", help="comma list of substrings of kernel names")
    args = ap.parse_args()
    if not torch.cuda.is_available():
        raise SystemExit("kernel_probe.py needs an MI355X (libsgf has no CPU path)")
    dtype = torch.float32 if args.dtype == "f32" else torch.bfloat16
    only = tuple(s for s in args.only.split(",") if s)
    for row in run(args.nodes, args.hidden, dtype, "cuda:0", args.reps, only):
        print(json.dumps({"n": args.nodes, "d": args.hidden, "dtype": args.dtype, **row}), flush=True)


if __name__ == "__main__":
    main()
