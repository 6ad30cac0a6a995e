#!/usr/bin/env python
"""torch.profiler view of one bench step (which ATen ops surround the libsgf kernels).
    python scripts/torch_profile_step.py [--dtype bf16] [--nodes N]
Prints the top GPU-time ops grouped by (name, input shapes)."""
import argparse
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sgformer_amd import synth  # noqa: E402
from sgformer_amd.ours import SGFormer  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--nodes", type=int, default=0)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    n, avg_deg, f, c, d = synth.SHAPES["ogbn-products"]
    n = args.nodes or n
    ei = synth.synthetic_graph(n, avg_deg, seed=123, device=dev)
    x, y, idx = synth.synthetic_task(n, f, c, seed=123)
    dt = torch.float32 if args.dtype == "f32" else torch.bfloat16
    x, y, idx = x.to(dev, dt), y.to(dev), idx.to(dev)
    model = SGFormer(f, d, c, trans_dropout=0.0, gnn_dropout=0.0, compute_dtype=None if args.dtype == "f32" else dt,
                     **synth.RECIPES["ogbn-products"]).to(dev).train()
    opt = torch.optim.Adam(model.parameters(), lr=0.01)

    def step():
        opt.zero_grad(set_to_none=True)
        out = model(x, ei)
        loss = F.nll_loss(F.log_softmax(out.float(), dim=1)[idx], y[idx])
        loss.backward()
        opt.step()

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        step()
        torch.cuda.synchronize()
    print(prof.key_averages(group_by_input_shape=True).table(sort_by="cuda_time_total", row_limit=int(os.environ.get("SGF_PROFILE_ROWS", "45")),
                                                             max_name_column_width=48, max_shapes_column_width=60))


if __name__ == "__main__":
    main()
