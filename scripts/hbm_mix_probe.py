#!/usr/bin/env python
"""What HBM gives a streaming kernel by read : write mix (ATen's elementwise kernels on 1.25 GB bf16 operands) — the
yardstick for the [N, d] kernels' `of copy rate` column."""
import json
import torch

dev = torch.device("cuda:0")
n, d = 2449029, 256
a = torch.randn(n, d, device=dev).to(torch.bfloat16)
b = torch.randn(n, d, device=dev).to(torch.bfloat16)
c = torch.randn(n, d, device=dev).to(torch.bfloat16)
o = torch.empty_like(a)
T = a.numel() * 2 / 1e9


def timed(fn, reps=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record()
    for i in range(reps):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(reps))
    return ts[len(ts) // 2]


res = {}
for name, fn, t in (("fill (0R:1W)", lambda: o.fill_(1.0), 1), ("sum (1R:0W)", lambda: a.sum(), 1),
                    ("copy (1R:1W)", lambda: o.copy_(a), 2), ("add (2R:1W)", lambda: torch.add(a, b, out=o), 3),
                    ("addcmul (3R:1W)", lambda: torch.addcmul(a, b, c, out=o), 4),
                    ("relu_ in place (1R:1W same lines)", lambda: a.relu_(), 2)):
    ms = timed(fn)
    res[name] = {"ms": round(ms, 4), "TB/s": round(t * T / ms, 3)}
print(json.dumps(res, indent=1))
