#!/usr/bin/env python
"""A/B of the node reductions: csrc/gramx.hip (tiles by LDS-DMA) against the register-staged k_reduce_bf16 (attn.hip),
interleaved in ONE process on the same tensors (SGF_GRAMX toggled through sgf_reload_env).

    python scripts/gramx_probe.py [--nodes 2449029] [--rounds 5]

One JSON line per case: median launch time of both arms (HIP events on the launch stream, finalize kernels included),
algorithmic bytes, GB/s and the fraction of the 6.3 TB/s a device copy reaches."""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sgformer_amd import _lib, ops  # noqa: E402


def switch(on):
    os.environ["SGF_GRAM_BN2"] = "1"        # (opt-in in the library; the layer case of this probe times it)
    os.environ["SGF_GRAMX"] = "1" if on else "0"
    _lib.load().sgf_reload_env()


def timed(fn, reps=5):
    fn()
    evs = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    return sorted(x.elapsed_time(y) for x, y in evs)[len(evs) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nodes", type=int, default=2449029)
    ap.add_argument("--hidden", type=int, default=256)
    ap.add_argument("--rounds", type=int, default=5)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    n, d = args.nodes, args.hidden
    K = ops.K
    g = torch.Generator().manual_seed(0)

    def act(w=d):
        return torch.randn(n, w, generator=g).to(dev, torch.bfloat16)

    h, gy, x0 = act(), act(), act()
    gp = act(48)
    rowscal = torch.rand(n, 2, generator=g).to(dev)
    dw = torch.empty(d, 2 * d, device=dev)
    xf = act(104)
    xs2 = act()
    mean, rstd = torch.randn(d, generator=g).to(dev) * 0.1, (1.0 + torch.rand(d, generator=g)).to(dev)
    gamma, beta = (1.0 + 0.1 * torch.randn(d, generator=g)).to(dev), (0.1 * torch.randn(d, generator=g)).to(dev)
    stats = torch.randn(2 * d, generator=g).to(dev)
    rmean, rrstd = torch.randn(n, generator=g).to(dev) * 0.1, (1.0 + torch.rand(n, generator=g)).to(dev)
    T = n * d * 2
    cases = [
        ("sgf_gram(h, h)  G", T, lambda: K.gram(h, h)),
        ("sgf_gram(gy, h)  dW", 2 * T, lambda: K.gram(gy, h)),
        ("sgf_gram(gp[n,48], h)  head dW", T + n * 96, lambda: K.gram(gp, h)),
        ("sgf_gram2(gy; h, x0)  paired", 3 * T, lambda: K.gram2(gy, h, x0, dw[:, :d], dw[:, d:])),
        ("sgf_attn_h_bwd_reduce_scaled", 2 * T + n * 8, lambda: K.attn_h_bwd_reduce_scaled(h, gy, rowscal)),
        ("sgf_gram_bn_bwd (g1, g2, z; x[n,104])", 3 * T + n * 208, lambda: K.gram_bn_bwd(h, gy, x0, mean, rstd, gamma, beta, True, stats, 1.0 / n, True, xf)),
        ("sgf_gram_ln_bwd (g, hpre; x[n,104])", 2 * T + n * 216, lambda: K.gram_ln_bwd(gy, h, rmean, rrstd, gamma, beta, True, xf)),
        ("bn_bwd_apply + gram2 | sgf_gram2_bn_bwd (g, z; y, x0)", 6 * T, None),
        ("copy (yardstick)", 2 * T, lambda: h.clone()),
    ]
    lstats = K.bn_bwd_stats(gy, h, mean, rstd, gamma, beta, True)

    def layer_old():
        dzz = K.bn_bwd_apply(gy, h, mean, rstd, gamma, beta, True, lstats, 1.0 / n, True)
        K.gram2(dzz, x0, xs2, dw[:, :d], dw[:, d:])

    def layer_new():
        K.gram2_bn_bwd(gy, h, mean, rstd, gamma, beta, True, lstats, 1.0 / n, True, x0, xs2, dw[:, :d], dw[:, d:])
    for name, nbytes, fn in cases:
        res = {"old": [], "new": []}
        for _ in range(args.rounds):
            for arm in ("old", "new"):
                switch(arm == "new" or fn is None)       # (the layer case: both arms on the DMA kernels, fused vs two launches)
                res[arm].append(timed(fn if fn is not None else (layer_new if arm == "new" else layer_old)))
        row = {"n": n, "d": d, "case": name, "algorithmic_GB": round(nbytes / 1e9, 3)}
        for arm in ("old", "new"):
            ms = sorted(res[arm])[len(res[arm]) // 2]
            row[arm + "_ms"] = round(ms, 4)
            row[arm + "_of_copy"] = round(nbytes / ms / 1e6 / 6300.0, 3)
        print(json.dumps(row), flush=True)
    switch(True)


if __name__ == "__main__":
    main()
