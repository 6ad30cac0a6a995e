#!/usr/bin/env python
"""bench.py — SGFormer fwd+bwd nodes/s on an ogbn-products-shaped synthetic graph (BASELINE.json).

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One step = one full-graph training step of the drop-in SGFormer (sgformer_amd/ours.py) with the
products recipe of large/run.sh:15-19 (hidden 256, 3 GCN layers with use_init, 1 attention layer,
dropout 0): forward, log_softmax + NLL on the training rows (large/main.py:139-141), backward, and
the reference's two-group Adam step (large/main.py:114-119).  Inputs are resident in HBM before the
timed region.  N > 1 shards the SAME graph by node ranges (strong scaling; sgformer_amd/dist.py).
`--workload papers100M-weak` is the weak-scaling variant of BASELINE.json config 5: 13.9 M nodes PER RANK
of a (13.9 M x N)-node uniform random graph (111 M nodes at N = 8, hidden 128, 100M/run.sh recipe); every
rank generates only its own rows (synth.synthetic_graph_shard) and no rank ever holds the global edge list.

Rank 0 prints ONE JSON line: the contract fields plus
  roofline      — the dominant kernel (CSR SpMM, k_spmm_wave): algorithmic bytes per launch (SURVEY.md
                  §8d: nnz*8 + (rows+1)*8 + X read once + Y written once) / mean launch time measured
                  with HIP events on the launch stream inside the timed region, against 8 TB/s;
                  `gather_bytes` is the no-reuse traffic of the same launch (each stored entry
                  fetching a d-wide row), the honest bound for a uniform random graph.
  cpu_baseline  — the CPU restatement of the reference (oracle/, torch CPU kernels, all host cores) on
                  a bounded sample of the same workload, timed on this box before the GPU run.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from sgformer_amd import ops, synth  # noqa: E402
from sgformer_amd.loss import log_softmax_nll  # noqa: E402
from sgformer_amd.dist import ShardContext, shard_model, sharded_nll_loss  # noqa: E402
from sgformer_amd.ours import SGFormer  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s measured copy ceiling)

# HBM bytes per k_spmm_wave launch from the rocprofv3 PMC passes committed under profiles/
# (separate --pmc FETCH_SIZE / WRITE_SIZE runs; FETCH_SIZE x2 per the gfx950 correction of
# MI355X_MICROARCH.md §HBM, which reproduces the gather bytes of this graph to 0.3 %).  PMC counters
# cannot be collected from inside this process, so the figure is keyed on the exact workload.
PMC_SPMM_TRAFFIC = {("ogbn-products", "bf16"): (67.1e9 + 1.25e9, "profiles/r01_spmm_pmc.md")}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="ogbn-products", choices=sorted(synth.SHAPES))
    ap.add_argument("--dtype", default="bf16", choices=["f32", "bf16"],
                    help="activation storage (BASELINE.json config 3 is bf16; f32 = the reference numerics)")
    ap.add_argument("--nodes", type=int, default=0, help="override N (debug only; marks the line)")
    ap.add_argument("--cpu-sample-nodes", type=int, default=200000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--seed", type=int, default=123)
    ap.add_argument("--aten-loss", action="store_true",
                    help="time the step with the trainer's own F.log_softmax + F.nll_loss (5 ATen kernels) "
                         "instead of sgformer_amd.loss.log_softmax_nll")
    ap.add_argument("--no-locality-probe", action="store_true",
                    help="skip the SpMM-only measurement on the locality-structured graph")
    return ap.parse_args()


def _cpu_step_time(workload, n, seed, threads, reps):
    from oracle import sgformer_oracle as O
    _, avg_deg, f, c, d = synth.SHAPES[workload]
    cfg = dict(synth.RECIPES.get(workload, synth.RECIPES["ogbn-products"]))
    torch.set_num_threads(threads)
    ei = synth.synthetic_graph(n, avg_deg, seed=seed)
    x, y, idx = synth.synthetic_task(n, f, c, seed=seed)
    p = O.init_params(cfg, f, d, c, seed=0)
    for k, v in p.items():
        if v.is_floating_point() and "running" not in k:
            v.requires_grad_(True)
    adj = O.build_adj(ei, n)
    times = []
    for _ in range(reps + 1):
        for v in p.values():
            v.grad = None
        t0 = time.perf_counter()
        loss = O.nll_loss(O.sgformer_forward(p, x, ei, cfg, training=True, adj=adj), y, idx)
        loss.backward()
        times.append(time.perf_counter() - t0)
    times = sorted(times[1:])          # first iteration is the warm-up
    return times[len(times) // 2], int(ei.shape[1])


def cpu_baseline(workload: str, n_sample: int, seed: int, budget_s: float = 20.0):
    """oracle/ (test infrastructure) used ONLY here, as the thing measured against — never as a
    fallback.  Same recipe, same average degree, fp32, dropout 0; CSR SpMM via torch.sparse (MKL)
    built once so the CPU is not handicapped.  torch's CPU kernels do not scale to
    every core of a 256-core host (a first run with 256 threads was 8x SLOWER than 8 threads), so a
    short probe picks the fastest thread count, and the sample size is cut so that the timed part
    stays within ~`budget_s` seconds (cost is linear in N and nnz)."""
    n_full = synth.SHAPES[workload][0]
    cores = os.cpu_count() or 1
    probe_n = min(20000, n_full)
    cands = sorted({t for t in (8, 16, 32, 64, 128, cores) if t <= cores})
    best_t, best = cands[0], float("inf")
    for t in cands:
        dt, _ = _cpu_step_time(workload, probe_n, seed, t, reps=1)
        if dt < best:
            best_t, best = t, dt
    n = int(min(n_sample, n_full, max(probe_n, probe_n * (budget_s / 4.0) / best)))
    dt, nnz = _cpu_step_time(workload, n, seed, best_t, reps=3)
    return {"value": n / dt, "unit": "nodes/s", "cores": best_t, "kind": "port",
            "sample": f"{workload}-shaped uniform random graph cut to N={n} (nnz={nnz}), same recipe, "
                      f"fp32, dropout 0, fwd+loss+bwd, median of 3 after 1 warm-up, {dt * 1e3:.0f} ms/step, "
                      f"{best_t} of {cores} host threads (fastest of {cands} in a {probe_n}-node probe)"}


class SpmmTimer:
    """HIP-event timing of every SpMM launch on the launch stream (torch's current stream)."""

    def __init__(self):
        self.pairs, self.bytes_alg, self.bytes_gather, self.active, self.d = [], [], [], False, 0
        self._orig = ops.K.spmm

    def install(self):
        timer, orig = self, self._orig

        def timed(rowptr, colind, val, x, n_rows, **kw):
            if not timer.active:
                return orig(rowptr, colind, val, x, n_rows, **kw)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            y = orig(rowptr, colind, val, x, n_rows, **kw)
            b.record()
            s = x.element_size()
            nnz, d = colind.numel(), x.shape[1]
            timer.d = d
            timer.pairs.append((a, b))
            timer.bytes_alg.append(nnz * 8 + (n_rows + 1) * 8 + x.shape[0] * d * s + n_rows * d * s)
            timer.bytes_gather.append(nnz * (8 + d * s) + (n_rows + 1) * 8 + n_rows * d * s)
            return y

        ops.K.spmm = timed

    def summary(self):
        if not self.pairs:
            return None
        ms = [a.elapsed_time(b) for a, b in self.pairs]
        mean_ms = sum(ms) / len(ms)
        alg = sum(self.bytes_alg) / len(self.bytes_alg)
        gat = sum(self.bytes_gather) / len(self.bytes_gather)
        achieved = alg / (mean_ms * 1e-3) / 1e9
        kern = "k_spmm_wave" if self.d > 128 else "k_spmm_sub"
        return {"kernel": f"{kern} (sgf_spmm)", "bound": "hbm", "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": None, "launches": len(ms), "mean_launch_ms": round(mean_ms, 4),
                "algorithmic_bytes": int(alg), "gather_bytes": int(gat),
                "gather_GBps": round(gat / (mean_ms * 1e-3) / 1e9, 1)}


def spmm_locality_probe(n, avg_deg, d, dtype, seed, dev, reps=5, locality=0.9, window=4096):
    """The SAME sgf_spmm kernel on a graph of the same size whose edges are mostly local in node id
    (synth.synthetic_graph_local).  On the uniform random graph of the headline workload every
    stored entry must fetch its 512-byte neighbour row from HBM (X is 1.25 GB: 5x the Infinity
    Cache, 300x an XCD's L2), so the kernel is bound by GATHER bytes, 19x the algorithmic bytes;
    this probe shows what the kernel does with the reuse a real graph offers."""
    ei = synth.synthetic_graph_local(n, avg_deg, locality=locality, window=window, seed=seed, device=dev)
    graph = ops.CSRGraph(ei, n, validate=False)
    nnz = int(ei.shape[1])
    del ei
    x = torch.randn(n, d, device=dev).to(dtype)
    for _ in range(2):
        ops.K.spmm(graph.rowptr, graph.colind, graph.val, x, n, long_segments=graph.long_segments)
    evs = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        ops.K.spmm(graph.rowptr, graph.colind, graph.val, x, n, long_segments=graph.long_segments)
        b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in evs)[len(evs) // 2]
    s = x.element_size()
    alg = nnz * 8 + (n + 1) * 8 + 2 * n * d * s
    return {"graph": f"same N / degree, {locality:.0%} of pairs within ~N(0,{window}) ids, rest uniform",
            "nnz": nnz, "launch_ms": round(ms, 4), "algorithmic_bytes": int(alg),
            "achieved": round(alg / (ms * 1e-3) / 1e9, 1), "unit": "GB/s",
            "frac": round(alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}


def make_inputs(workload: str, nodes: int, seed: int, rank: int, world: int, dev):
    """Synthetic inputs of one rank (host x / y / train_idx, edge_index on `dev`) and its ShardContext.
    Strong-scaling workloads: every rank generates the SAME global graph and task and keeps its rows.
    `*-weak`: SHAPES gives the node count PER RANK; the rank generates only its own rows of the
    (n_per * world)-node graph (global ids), its own features / labels / split."""
    n, avg_deg, f, c, d = synth.SHAPES[workload]
    if nodes:
        n = nodes
    cfg = dict(synth.RECIPES.get(workload, synth.RECIPES["ogbn-products"]))
    weak = workload.endswith("-weak")
    ctx = None
    if weak:
        n_per, n = n, n * world
        ei = synth.synthetic_graph_shard(n_per, avg_deg, rank, world, seed=seed, device=dev)
        x, y, train_idx = synth.synthetic_task(n_per, f, c, seed=seed + 7919 * rank)
        n_train = train_idx.numel() * world
        if world > 1:
            ctx = ShardContext(n, local_edges=True)
    else:
        ei = synth.synthetic_graph(n, avg_deg, seed=seed, device=dev)
        x, y, train_idx = synth.synthetic_task(n, f, c, seed=seed)
        n_train = train_idx.numel()
        if world > 1:
            ctx = ShardContext(n)
            x, y, train_idx = ctx.shard_rows(x), ctx.shard_rows(y), ctx.local_index(train_idx)
    return n, f, c, d, cfg, weak, ei, x, y, train_idx, n_train, ctx


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (see docstring)")
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback by design)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args.workload, args.cpu_sample_nodes, args.seed)

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    n, f, c, d, cfg, weak, ei, x, y, train_idx, n_train, ctx = make_inputs(args.workload, args.nodes, args.seed,
                                                                           rank, world, dev)
    dtype = torch.float32 if args.dtype == "f32" else torch.bfloat16
    x, y, train_idx = x.to(dev, dtype), y.to(dev), train_idx.to(dev)

    torch.manual_seed(args.seed)
    # bf16 = bf16 activation storage with fp32 master weights and fp32 accumulation everywhere
    model = SGFormer(f, d, c, trans_dropout=0.0, gnn_dropout=0.0,
                     compute_dtype=None if args.dtype == "f32" else dtype, **cfg).to(dev)
    if ctx is not None:
        shard_model(model, ctx)
    opt = torch.optim.Adam([{"params": model.params1, "weight_decay": 1e-5},
                            {"params": model.params2, "weight_decay": 1e-5}], lr=0.01)
    model.train()

    aten_loss = args.aten_loss

    def step():
        opt.zero_grad(set_to_none=True)
        logits = model(x, ei)
        if ctx is not None:
            loss = sharded_nll_loss(logits, y, train_idx, n_train)
        elif aten_loss:   # the three lines of large/main.py:139-141 as the trainer writes them
            loss = F.nll_loss(F.log_softmax(logits.float(), dim=1)[train_idx], y[train_idx])
        else:             # the same arithmetic in one pass (sgf_nll_fwd / sgf_nll_bwd)
            loss = log_softmax_nll(logits, y, train_idx)
        loss.backward()
        if ctx is not None:
            ctx.sync_grads(model.parameters())
        opt.step()
        return loss

    timer = SpmmTimer()
    timer.install()
    for _ in range(args.warmup):
        step()

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    fence()
    timer.active = True
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    fence()
    elapsed = time.perf_counter() - t0
    timer.active = False   # (the extra ATen-loss steps below are not part of the roofline sample)
    loss_val = float(loss.detach())
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)
        lt = torch.tensor([loss_val], device=dev, dtype=torch.float64)
        dist.all_reduce(lt)
        loss_val = float(lt)

    ms_aten = None
    if world == 1 and not aten_loss:
        # transparency: the same step with the trainer's own ATen loss ops
        aten_loss = True
        step()
        fence()
        t1 = time.perf_counter()
        for _ in range(min(args.steps, 5)):
            step()
        fence()
        ms_aten = (time.perf_counter() - t1) / min(args.steps, 5) * 1e3
        aten_loss = False
    roof = timer.summary()
    if roof is not None and world == 1 and not args.nodes and (args.workload, args.dtype) in PMC_SPMM_TRAFFIC:
        roof["traffic"], roof["traffic_source"] = PMC_SPMM_TRAFFIC[(args.workload, args.dtype)]
    if rank == 0 and world == 1 and roof is not None and not args.no_locality_probe and not args.nodes:
        del model, opt, x, y, loss
        ops.graph_cache.clear()
        torch.cuda.empty_cache()
        roof["locality_probe"] = spmm_locality_probe(n, synth.SHAPES[args.workload][1], d, dtype, args.seed, dev)

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        line = {
            "metric": f"SGFormer fwd+bwd nodes/sec on {args.workload} full-graph",
            "value": n * args.steps / elapsed, "unit": "nodes/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True,
            "scaling": "weak" if weak else "strong", "vs_baseline": None, "dtype": args.dtype,
            "data": "synthetic",
            "config": {"workload": f"{args.workload}-shaped uniform random graph, full-graph "
                                   f"training step (fwd + log_softmax/NLL on the training rows + bwd + Adam), "
                                   f"{'100M' if 'papers' in args.workload else 'large'}/run.sh recipe, dropout 0"
                                   + (f"; {n // world:,} nodes per rank, rows generated per rank" if weak else ""),
                       "loss": "F.log_softmax + F.nll_loss (ATen, as large/main.py:139-141 writes it)" if args.aten_loss
                               else "sgformer_amd.loss.log_softmax_nll (same arithmetic, one pass)",
                       "ms_per_step_with_aten_loss": None if ms_aten is None else round(ms_aten, 3),
                       "nodes": n, ("nnz_per_rank" if weak else "nnz"): int(ei.shape[1]), "features": f, "hidden": d, "classes": c,
                       "parallelism": f"node-shard x{world}" if world > 1 else "single GPU",
                       "debug_override": bool(args.nodes)},
            "loss": loss_val,
            "peak_mem_GB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2),
            "roofline": roof,
            "cpu_baseline": cpu,
        }
        if cpu is not None:
            line["speedup_vs_cpu_baseline"] = round(line["value"] / cpu["value"], 1)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
