#!/usr/bin/env python
"""One epoch of the reference's random-partition mini-batch loop (large/main-batch.py:134-151) at
ogbn-products scale with the amazon2m recipe (large/run.sh:15-19: batch_size 100000), run the way
the unchanged trainer runs it under sgformer_amd.launch: host feature gather + H2D per batch, GPU
induced subgraph (batching.subgraph), model step on the GPU.
    python scripts/minibatch_probe.py [--dtype f32] [--batch 100000]"""
import argparse
import json
import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sgformer_amd import batching, synth  # noqa: E402
from sgformer_amd.ours import SGFormer  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"])
    ap.add_argument("--batch", type=int, default=100000)
    ap.add_argument("--host-features", action="store_true",
                    help="keep node features on the host (the reference trainer as written; default = what "
                         "sgformer_amd.launch sets up: features resident on the GPU, 16 host threads)")
    args = ap.parse_args()
    if not args.host_features:
        from sgformer_amd import launch
        launch.limit_host_threads()
    dev = torch.device("cuda:0")
    n, avg_deg, f, c, d = synth.SHAPES["ogbn-products"]
    ei = synth.synthetic_graph(n, avg_deg, seed=123, device=dev).cpu()     # data stays on the HOST (main-batch.py:43-99)
    x, y, train_idx = synth.synthetic_task(n, f, c, seed=123)
    if not args.host_features:
        x = x.to(dev)                      # launch.patch_resident_features
    train_mask = torch.zeros(n, dtype=torch.bool)
    train_mask[train_idx] = True
    dt = None if args.dtype == "f32" else torch.bfloat16
    model = SGFormer(f, d, c, trans_dropout=0.0, gnn_dropout=0.0, compute_dtype=dt,
                     **synth.RECIPES["ogbn-products"]).to(dev)
    opt = torch.optim.Adam([{"params": model.params1}, {"params": model.params2}], lr=0.01)
    num_batch = n // args.batch + (n % args.batch > 0)

    def epoch():
        model.train()
        idx = torch.randperm(n)
        t_sub = t_gather = 0.0
        for i in range(num_batch):
            idx_i = idx[i * args.batch:(i + 1) * args.batch]
            t0 = time.perf_counter()
            train_mask_i = train_mask[idx_i]
            x_i = x[idx_i].to(dev)
            t1 = time.perf_counter()
            ei_i, _ = batching.subgraph(idx_i, ei, num_nodes=n, relabel_nodes=True)
            ei_i = ei_i.to(dev)
            t2 = time.perf_counter()
            y_i = y[idx_i].to(dev)
            opt.zero_grad()
            out = F.log_softmax(model(x_i, ei_i), dim=1)
            loss = F.nll_loss(out[train_mask_i], y_i[train_mask_i])
            loss.backward()
            opt.step()
            t_gather += t1 - t0
            t_sub += t2 - t1
        torch.cuda.synchronize()
        return t_gather, t_sub, float(loss)

    epoch()
    t0 = time.perf_counter()
    tg, ts, loss = epoch()
    dt_epoch = time.perf_counter() - t0
    print(json.dumps({"workload": "ogbn-products-shaped, amazon2m recipe, random-partition mini-batches",
                      "features": "host" if args.host_features else "resident on the GPU", "host_threads": torch.get_num_threads(),
                      "dtype": args.dtype, "batch_nodes": args.batch, "batches": num_batch,
                      "epoch_s": round(dt_epoch, 3), "nodes_per_s": round(n / dt_epoch),
                      "host_gather_h2d_s": round(tg, 3), "gpu_subgraph_s": round(ts, 3),
                      "model_and_rest_s": round(dt_epoch - tg - ts, 3), "loss": loss}))


if __name__ == "__main__":
    main()
