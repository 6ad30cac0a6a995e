#!/usr/bin/env python
"""The dense half of one GraphConv layer (large/ours.py:36-40, 87-93) at the ogbn-products shape, forward and backward:
the round-4 kernels (one-pass two-operand Linear, fused BatchNorm-backward + both input gradients with the running x0
gradient, paired Gram) against the round-3 sequence of launches they replace.  Interleaved timings, median of 12."""
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
    return round(ts[len(ts) // 2], 4)


def main():
    dev = torch.device("cuda:0")
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2449029
    d = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    g = torch.Generator(device=dev).manual_seed(1)
    K = ops.K
    y = torch.randn(n, d, device=dev, generator=g).bfloat16()
    x0 = torch.randn(n, d, device=dev, generator=g).bfloat16()
    gy = torch.randn(n, d, device=dev, generator=g).bfloat16()
    w = (torch.randn(d, 2 * d, device=dev, generator=g) / (2 * d) ** 0.5).bfloat16()
    bias = torch.randn(d, device=dev, generator=g)
    shift = torch.randn(d, device=dev, generator=g) * 0.1
    mean = torch.randn(d, device=dev, generator=g) * 0.2
    rstd = 1.0 / (1.0 + torch.rand(d, device=dev, generator=g))
    gamma = 1.0 + 0.3 * torch.randn(d, device=dev, generator=g)
    beta = 0.2 * torch.randn(d, device=dev, generator=g)
    t_gb = n * d * 2 / 1e9
    out = {"n": n, "d": d, "T_GB": round(t_gb, 4), "ms": {}}
    ms = out["ms"]

    # ---- forward: z = [y | x0] W^T + b with BatchNorm's sums ----
    def fwd(mode):
        os.environ["SGF_GCN_CAT"] = mode
        return K.gcn_epilogue_cat(y, x0, w, bias, shift, want_stats=True)
    z1, st1 = fwd("1")
    z0, st0 = fwd("0")
    out["cat one-pass vs two-pass max |dz|"] = float((z1.float() - z0.float()).abs().max())
    for rep in range(2):
        ms[f"fwd two-pass (partial + stats_add) #{rep}"] = timed(lambda: fwd("0"))
        ms[f"fwd one-pass paired (sgf_gcn_epilogue_cat) #{rep}"] = timed(lambda: fwd("1"))
    os.environ["SGF_GCN_CAT"] = "1"
    z = z1
    del z0, z1

    # ---- backward ----
    stats = K.bn_bwd_stats(gy, z, mean, rstd, gamma, beta, True)
    inv_n = 1.0 / n
    acc0 = K.gcn_bn_bwd_dx(gy, z, mean, rstd, gamma, beta, True, stats, inv_n, True, w, None, last=False, add_gy=True)[2]

    def old_bwd():
        dz = K.bn_bwd_apply(gy, z, mean, rstd, gamma, beta, True, stats, inv_n, True)
        dy = K.gcn_epilogue_dx(dz, w[:, :d])
        dx0 = K.gcn_epilogue_dx(dz, w[:, d:])
        return dz, dy, dx0

    def new_bwd(acc_in, last):
        return K.gcn_bn_bwd_dx(gy, z, mean, rstd, gamma, beta, True, stats, inv_n, True, w, acc_in, last=last, add_gy=True)

    dz_o, dy_o, dx0_o = old_bwd()
    dz_n, dy_n, tot = new_bwd(None, True)
    out["bwd fused vs separate: dz equal fraction"] = float((dz_o == dz_n).float().mean())
    out["bwd fused vs separate: max |d dy|"] = float((dy_o.float() - dy_n.float()).abs().max())
    out["bwd fused vs separate: max |d (dx0 + gy)|"] = float(((dx0_o.float() + gy.float()) - tot.float()).abs().max())
    del dz_o, dy_o, dx0_o, dz_n, dy_n, tot
    for rep in range(2):
        ms[f"bwd bn_bwd_stats #{rep}"] = timed(lambda: K.bn_bwd_stats(gy, z, mean, rstd, gamma, beta, True))
        ms[f"bwd bn_bwd_apply + 2 x dx #{rep}"] = timed(old_bwd)
        ms[f"bwd fused first (no running sum in) #{rep}"] = timed(lambda: new_bwd(None, False))
        ms[f"bwd fused middle (running sum in and out) #{rep}"] = timed(lambda: new_bwd(acc0, False))
        ms[f"bwd fused last (row-major total out) #{rep}"] = timed(lambda: new_bwd(acc0, True))
    xs7 = [gy, gy, gy, y, y, y, x0]
    ms["hub: sum of 7 gradients (k_sum_n)"] = timed(lambda: K.sum_n(xs7))

    # ---- weight gradients ----
    dz = K.bn_bwd_apply(gy, z, mean, rstd, gamma, beta, True, stats, inv_n, True)
    dw = torch.empty(d, 2 * d, device=dev)

    def two_grams():
        K.gram(dz, y, out=dw[:, :d], want_colsum=True)
        K.gram(dz, x0, out=dw[:, d:], want_colsum=False)
    for rep in range(2):
        ms[f"dW two sgf_gram #{rep}"] = timed(two_grams)
        ms[f"dW paired sgf_gram2 #{rep}"] = timed(lambda: K.gram2(dz, y, x0, dw[:, :d], dw[:, d:], want_colsum=True))
    out["TBps"] = {
        "fwd one-pass on 3T": round(3 * t_gb / ms["fwd one-pass paired (sgf_gcn_epilogue_cat) #1"], 3),
        "bwd fused middle on 6T": round(6 * t_gb / ms["bwd fused middle (running sum in and out) #1"], 3),
        "gram2 on 3T": round(3 * t_gb / ms["dW paired sgf_gram2 #1"], 3),
    }
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
