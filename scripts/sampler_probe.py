#!/usr/bin/env python
"""The 100M recipe's training loop (100M/nb-sample.py:169-175: NeighborLoader [15, 10, 5], batch_size seeds,
model(graph.x, graph.edge_index)[:batch_size], CrossEntropyLoss, Adam) with the DEVICE sampler
(sgformer_amd.sampling.NeighborLoader) on the share of a papers100M-shaped graph one of 8 GPUs holds
(synth.SHAPES['papers100M-shard8']: 13.9 M nodes, ~416 M stored entries, 128 features, 172 classes, hidden 128).
Prints one JSON line: ms per sampled batch (sampling alone / with the feature gather / with the model step),
sampled nodes and edges per batch, seeds per second.
    python scripts/sampler_probe.py [--nodes N] [--batch 1000] [--batches 30] [--dtype bf16]"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sgformer_amd import synth  # noqa: E402
from sgformer_amd.ours_100m import SGFormer  # noqa: E402
from sgformer_amd.sampling import NeighborLoader  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nodes", type=int, default=0)
    ap.add_argument("--batch", type=int, default=1000)
    ap.add_argument("--batches", type=int, default=30)
    ap.add_argument("--dtype", default="bf16", choices=["f32", "bf16"])
    ap.add_argument("--scale", default="shard8", choices=["shard8", "papers100M"],
                    help="papers100M: the WHOLE graph's size on one GPU — 111 059 956 nodes, 1 615 685 872 stored entries with a "
                         "heavy-tailed in-degree (hubs of ~1e5 entries), features in bf16 (28 GB), everything generated on the device")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    n, deg, f, c, d = synth.SHAPES["papers100M-shard8"]
    n = a.nodes or n
    t_gen = time.perf_counter()
    if a.scale == "papers100M":
        n = a.nodes or 111_059_956
        m = int(1_615_685_872 * (n / 111_059_956))
        g = torch.Generator(device=dev).manual_seed(123)
        src = torch.randint(0, n, (m,), device=dev, generator=g)
        # targets ~ n * u^2: in-degree density 1 / (2 sqrt(v / n)) — node 0 collects ~ m / sqrt(n) entries
        dst = (torch.rand(m, device=dev, generator=g, dtype=torch.float64).square_() * n).long().clamp_(max=n - 1)
        key = torch.unique(dst * n + src)                # coalesce (sorted by target): stored entries are distinct
        del src, dst
        ei = torch.stack([key % n, key // n])            # [source, target]
        del key
        x = torch.empty((n, f), dtype=torch.bfloat16 if a.dtype == "bf16" else torch.float32, device=dev)
        for lo in range(0, n, 1 << 22):
            x[lo:lo + (1 << 22)] = torch.randn((min(1 << 22, n - lo), f), device=dev, generator=g).to(x.dtype)
        y = torch.randint(0, c, (n,), device=dev, generator=g)
    else:
        ei = synth.synthetic_graph(n, deg, seed=123, device=dev)
        x, y, _ = synth.synthetic_task(n, f, c, seed=123, device=dev)
    torch.cuda.synchronize()
    t_gen = time.perf_counter() - t_gen

    class Data:
        pass
    data = Data()
    data.x, data.y, data.edge_index = x, y, ei
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    loader = NeighborLoader(data, input_nodes=torch.arange(0, a.batch * a.batches), num_neighbors=[15, 10, 5],
                            batch_size=a.batch, shuffle=True, seed=7,
                            feature_dtype=None if a.dtype == "f32" else torch.bfloat16)
    torch.cuda.synchronize()
    t_build = time.perf_counter() - t0
    del ei
    dt = None if a.dtype == "f32" else torch.bfloat16
    model = SGFormer(f, d, c, trans_dropout=0.0, gnn_dropout=0.0, compute_dtype=dt,
                     **synth.RECIPES["papers100M-shard8"]).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=0.01)
    loss_fn = torch.nn.CrossEntropyLoss()

    def run(step: bool, gather: bool):
        torch.cuda.synchronize()
        t = time.perf_counter()
        nodes = edges = 0
        if not gather:          # sampling alone
            ids = loader.input_nodes
            for b in range(len(loader)):
                n_id, e, bs = loader.sampler.sample(ids[b * a.batch:(b + 1) * a.batch])
                nodes += n_id.numel()
                edges += e.shape[1]
        else:
            for g in loader:
                nodes += g.n_id.numel()
                edges += g.edge_index.shape[1]
                if step:
                    out = model(g.x, g.edge_index)[:g.batch_size]
                    loss = loss_fn(out.float(), g.y[:g.batch_size])
                    opt.zero_grad(set_to_none=True)
                    loss.backward()
                    opt.step()
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / len(loader) * 1e3, nodes / len(loader), edges / len(loader)

    run(True, True)                                   # warm-up (graph cache, allocator)
    ms_sample, nn_, ne_ = run(False, False)
    reads = loader.sampler.host_reads
    run(False, False)
    reads = (loader.sampler.host_reads - reads) / len(loader)
    fan_dev, loader.sampler._fan_dev = loader.sampler._fan_dev, None       # the hop-by-hop path (two host reads per hop)
    ms_sample_hops, _, _ = run(False, False)
    loader.sampler._fan_dev = fan_dev
    ms_gather, _, _ = run(False, True)
    ms_step, _, _ = run(True, True)
    kind = "uniform random" if a.scale == "shard8" else "heavy-tailed in-degree (targets ~ n u^2), max in-degree " + \
        str(loader.sampler.max_deg)
    print(json.dumps({"graph": f"{kind}, {n} nodes, {loader.sampler.colind.numel()} stored entries",
                      "generate_s": round(t_gen, 2), "host_reads_per_batch": reads,
                      "ms_per_batch_sampling_hop_by_hop": round(ms_sample_hops, 3),
                      "fanouts": [15, 10, 5], "seeds_per_batch": a.batch, "dtype": a.dtype,
                      "sampler_build_s": round(t_build, 2), "nodes_per_batch": round(nn_), "edges_per_batch": round(ne_),
                      "ms_per_batch_sampling": round(ms_sample, 3), "ms_per_batch_sampling_plus_gather": round(ms_gather, 3),
                      "ms_per_batch_with_model_step": round(ms_step, 3),
                      "seeds_per_s_training": round(a.batch / ms_step * 1e3),
                      "sampled_nodes_per_s_training": round(nn_ / ms_step * 1e3),
                      "peak_mem_GB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)}))


if __name__ == "__main__":
    main()
