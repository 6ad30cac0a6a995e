#!/usr/bin/env python
"""A/B of the BatchNorm element-wise kernels on bf16 rows: 8 bytes per lane (SGF_EW8=0) vs 16 (default), interleaved in ONE
process on the same tensors at ogbn-products size; the outputs are compared bit for bit (statistics: relative)."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sgformer_amd import _lib, ops  # noqa: E402

K = ops.K


def setenv(v):
    os.environ["SGF_EW8"] = str(v)
    _lib.load().sgf_reload_env()


def timed(fn, reps=15, warm=6):
    ts = []
    for i in range(reps + warm):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        if i >= warm:
            ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2]


def main():
    dev = torch.device("cuda:0")
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2449029
    d = 256
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.randn(n, d, device=dev, generator=g).to(torch.bfloat16)
    gy = torch.randn(n, d, device=dev, generator=g).to(torch.bfloat16)
    gy2 = torch.randn(n, d, device=dev, generator=g).to(torch.bfloat16)
    res = torch.randn(n, d, device=dev, generator=g).to(torch.bfloat16)
    mean = torch.randn(d, device=dev, generator=g) * 0.1
    rstd = torch.rand(d, device=dev, generator=g) + 0.5
    gamma = torch.rand(d, device=dev, generator=g) + 0.5
    beta = torch.randn(d, device=dev, generator=g) * 0.1
    T = n * d * 2 / 1e9
    cases = {
        "bn_apply + residual (2R:1W)": (lambda: K.bn_apply(x, mean, rstd, gamma, beta, res, True), 3 * T),
        "bn_apply (1R:1W)": (lambda: K.bn_apply(x, mean, rstd, gamma, beta, None, True), 2 * T),
        "bn_bwd_stats (2R)": (lambda: K.bn_bwd_stats(gy, x, mean, rstd, gamma, beta, True), 2 * T),
        "bn_bwd_stats2 (3R)": (lambda: K.bn_bwd_stats2(gy, gy2, x, mean, rstd, gamma, beta, True), 3 * T),
    }
    setenv(1)
    stats = K.bn_bwd_stats(gy, x, mean, rstd, gamma, beta, True)
    cases["bn_bwd_apply (2R:1W)"] = (lambda: K.bn_bwd_apply(gy, x, mean, rstd, gamma, beta, True, stats, 1.0 / n, True), 3 * T)
    y_ln, mu_ln, rs_ln = K.ln_fwd(x, res, 0.5, 0.5, gamma, beta, True, 1e-5)
    cases["ln_bwd, shared dx (4R:1W)"] = (lambda: K.ln_bwd(gy, y_ln, x, res, 0.5, 0.5, gamma, True, mu_ln, rs_ln)[0], 5 * T)
    out = []
    for name, (fn, gb) in cases.items():
        row = {"case": name}
        outs = {}
        for rnd in range(2):
            for v in (0, 1):
                setenv(v)
                ms = timed(fn)
                row[f"ms_ew8={v}_r{rnd}"] = round(ms, 4)
                row[f"TBps_ew8={v}_r{rnd}"] = round(gb / ms, 3)
                outs[v] = fn()
        a, b = outs[0].float(), outs[1].float()
        row["identical"] = bool(torch.equal(outs[0], outs[1]))
        row["max_rel"] = float(((a - b).abs() / a.abs().clamp_min(1e-3)).max())
        out.append(row)
        print(json.dumps(row), flush=True)
    setenv(1)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "ew8_probe.jsonl"), "a") as f:
        for r in out:
            f.write(json.dumps(r) + "\n")


if __name__ == "__main__":
    main()
