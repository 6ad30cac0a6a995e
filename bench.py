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
  roofline      — the dominant kernel (CSR SpMM): algorithmic bytes per launch (nnz*8 +
                  (rows+1)*8 + X read once + Y written once) / mean launch time measured with HIP events on
                  the launch stream inside the timed region, against 8 TB/s; `gather_bytes` is the no-reuse
                  traffic of the same launch (each stored entry fetching a d-wide row), the bound for a
                  uniform random graph (profiles/r02_gather_probe.md).  `traffic` = HBM bytes per launch
                  from the rocprofv3 PMC passes of THIS kernel on THIS workload (profiles/r06_spmm_pmc.json,
                  written by scripts/pmc_passes.sh; null when no pass has been recorded).
  structured    — the same training step and the same SpMM roofline on a graph of the same size WITH
                  community structure and RANDOMLY PERMUTED node ids (synth.synthetic_graph_community): the
                  locality has to be recovered by sgf_reorder and is then exploited by the LDS-staged
                  row-block kernel.  The uniform headline graph is an expander (nothing to recover).
                  `structured.powerlaw`: three steps of the same on a power-law community graph with global hubs
                  (synth.synthetic_graph_community_powerlaw: the long-row path and skewed communities).
  cpu_baseline  — the CPU restatement of the reference (oracle/, torch CPU kernels, all host cores) on
                  a bounded sample of the same workload, timed on this box before the GPU run.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

# multi-process GPU work on this pool needs dmabuf IPC (RCCL fails with `hipIpcGetMemHandle: invalid argument` otherwise); the
# boxes export it already — kept here so that a launcher with a scrubbed environment still works
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402
import torch.distributed as dist
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from sgformer_amd import ops, synth  # noqa: E402,F401  (tests/bench_modes.py reaches ops through this module)
from benchlib.cpu import cpu_baseline  # noqa: E402
from benchlib.model import per_rank_memory_model, scaling_model  # noqa: E402
from benchlib.timers import SpmmTimer, pmc_traffic, spmm_source_sha16, step_roofline  # noqa: E402,F401  (scripts/pmc_summarise.py reads the hash here)
from benchlib.workloads import _sharded, make_inputs, run_minibatch, run_workload  # noqa: E402,F401

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
    ap.add_argument("--loss", default="trainer", choices=["trainer", "fused", "aten"],
                    help="what the HEADLINE step computes its loss with: 'trainer' (default) = the three loss lines of "
                         "large/main.py:139-141 exactly as an unchanged trainer runs them under sgformer_amd.launch "
                         "(nn.NLLLoss served by the gather form, launch.patch_nll_loss); 'fused' = "
                         "sgformer_amd.loss.log_softmax_nll (same arithmetic, one pass); 'aten' = the same three lines "
                         "on ATen's own nll_loss kernels.  The other two are timed on a few extra steps and reported "
                         "in config.")
    ap.add_argument("--aten-loss", action="store_true", help="(older spelling of --loss aten)")
    ap.add_argument("--no-structured", action="store_true",
                    help="skip the second measurement on the community-structured graph with shuffled node ids")
    ap.add_argument("--graph", default="uniform", choices=["uniform", "community", "powerlaw", "rmat"],
                    help="graph generator of the HEADLINE measurement (default: uniform random, the r01 workload); rmat = "
                         "R-MAT with the Graph500 parameters (a, b, c = 0.57, 0.19, 0.19), ids shuffled")
    ap.add_argument("--dropout", default="0", choices=["0", "recipe"],
                    help="'recipe': the dropout probabilities of the workload's run.sh recipe (ogbn-arxiv: 0.5 / 0.5, "
                         "large/run.sh:2-5; the products / pokec recipes use 0) on sgf_dropout (Philox keyed on seed and "
                         "element index, mask recomputed in the backward) inside the timed step; '0': the parity setting")
    ap.add_argument("--mode", default="fullgraph", choices=["fullgraph", "minibatch"],
                    help="minibatch: one step = one EPOCH of the reference's random-partition mini-batch loop "
                         "(large/main-batch.py:129-151, batch_size 100000 as in large/run.sh:15-19) as the unchanged trainer "
                         "runs it under sgformer_amd.launch; value = N / epoch time")
    ap.add_argument("--batch-size", type=int, default=100000)
    ap.add_argument("--no-minibatch-leg", action="store_true",
                    help="skip the mini-batch epoch appended to `structured` (ogbn-products, N = 1)")
    return ap.parse_args()


def legs_summary(ms: float, roof0, structured):
    """One flat object with what every leg of the default run measured — step time, the SpMM's roofline fraction on algorithmic
    bytes, its measured HBM traffic and the whole-step HBM fraction — so that a reader of the line does not have to walk
    `structured` (the legs' full objects stay there)."""
    def one(ms_, roof):
        roof = roof or {}
        return {"ms_per_step": round(ms_, 3), "spmm_kernel": (roof.get("kernel") or "").split(" ")[0] or None,
                "spmm_launch_ms": roof.get("mean_launch_ms"), "spmm_frac": roof.get("frac"),
                "spmm_traffic_bytes": roof.get("traffic"), "step_frac": (roof.get("step") or {}).get("frac")}
    out = {"headline": one(ms, roof0)}
    if structured:
        out["community"] = one(structured["ms_per_step"], structured["roofline"])
        for k in ("powerlaw", "rmat", "fp32"):
            if k in structured:
                out[k if k != "fp32" else "uniform_fp32"] = one(structured[k]["ms_per_step"], structured[k]["roofline"])
        if "minibatch_epoch" in structured:
            out["minibatch_epoch"] = {"ms_per_epoch": structured["minibatch_epoch"]["ms_per_epoch"],
                                      "nodes_per_s": round(structured["minibatch_epoch"]["value"])}
    return out


DRYRUN = os.environ.get("SGF_BENCH_DRYRUN") == "1"


SHARE_GPU = os.environ.get("SGF_BENCH_SHARE_GPU") == "1"


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (see docstring)")
        args.gpus = world
    if DRYRUN:      # test-only mode (tests/test_dist.py): implemented under tests/, not here
        from tests.bench_modes import enter_dryrun
        enter_dryrun(sys.modules[__name__])
        if not args.nodes:
            raise SystemExit("SGF_BENCH_DRYRUN=1 needs --nodes (a few thousand)")
        dev = torch.device("cpu")
    else:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X (no CPU fallback by design)")
        if SHARE_GPU:   # validation-only mode on a 1-GPU box: implemented under tests/, not here
            local_rank = 0
            from tests.bench_modes import enter_shared_gpu
            enter_shared_gpu()
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not DRYRUN:
        cpu = cpu_baseline(args.workload, args.cpu_sample_nodes, args.seed)

    if _sharded(world):
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if DRYRUN or SHARE_GPU:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    if args.mode == "minibatch":
        if world != 1 or DRYRUN:
            raise SystemExit("--mode minibatch is a single-GPU measurement")
        steps, warmup = max(1, min(args.steps, 5)), max(2, min(args.warmup, 2))     # (2: every batch size captured before the clock starts)
        r = run_minibatch(args, dev, steps, warmup)
        ms = r["elapsed"] / steps * 1e3
        line = {"metric": f"SGFormer fwd+bwd nodes/sec on {args.workload}, random-partition mini-batch epoch "
                          f"(large/main-batch.py:129-151)",
                "value": r["n"] * steps / r["elapsed"], "unit": "nodes/s", "n_gpus": 1, "steps": steps, "warmup": warmup,
                "ms_per_step": ms, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": args.dtype,
                "data": "synthetic",
                "config": {"workload": f"{args.workload}-shaped {args.graph} graph, one step = one EPOCH of "
                                       f"{r['breakdown']['batches_per_epoch']} random partitions of {args.batch_size} nodes "
                                       f"(large/run.sh:15-19), the trainer's loop lines verbatim, features resident on the "
                                       f"GPU, labels / masks on the host, dropout 0",
                           "nodes": r["n"], "nnz": r["nnz"], "features": r["f"], "hidden": r["d"], "classes": r["c"],
                           "parallelism": "single GPU", "debug_override": bool(args.nodes)},
                "loss": r["loss"], "peak_mem_GB": r["peak_mem"], "minibatch": r["breakdown"], "roofline": r["roof"],
                "cpu_baseline": cpu}
        print(json.dumps(line), flush=True)
        return
    r = run_workload(args, args.graph, rank, world, dev, args.steps, args.warmup, with_aten=True)
    structured = None
    if (rank == 0 and world == 1 and args.graph == "uniform" and not args.no_structured
            and not args.workload.endswith("-weak") and args.workload != "cora"):
        k = max(3, min(args.steps, 5))
        q = run_workload(args, "community", rank, world, dev, k, 3)
        structured = {
            "graph": "same N / degree; communities of 64-256 nodes in super-communities of 64 (80 % / 15 % / 5 % of "
                     "the pairs inside the community / the super-community / anywhere), node ids randomly permuted",
            "nnz": q["nnz"], "value": q["n"] * k / q["elapsed"], "unit": "nodes/s", "steps": k,
            "ms_per_step": round(q["elapsed"] / k * 1e3, 3), "loss": q["loss"], "graph_view": q["view"],
            "prepare_graph_s": None if q["prepare_s"] is None else round(q["prepare_s"], 3),
            "roofline": {**q["roof"], "step": step_roofline(q["n"], q["nnz"], q["f"], q["d"], q["cfg"], args.dtype,
                                                           q["elapsed"] / k * 1e3)}}
        # the same once more with a power-law structure: community sizes 16-4096 (truncated Pareto), heavy-tailed
        # endpoints inside a community, 3 % of the pairs to global hubs (rows of tens of thousands of entries)
        q = run_workload(args, "powerlaw", rank, world, dev, 3, 1)
        structured["powerlaw"] = {
            "graph": "same N / degree; community sizes 16-4096 by a truncated Pareto law, local hubs, 3 % of the pairs "
                     "to global hubs, node ids randomly permuted (synth.synthetic_graph_community_powerlaw)",
            "nnz": q["nnz"], "value": q["n"] * 3 / q["elapsed"], "unit": "nodes/s", "steps": 3,
            "ms_per_step": round(q["elapsed"] / 3 * 1e3, 3), "graph_view": q["view"],
            "roofline": {**q["roof"], "step": step_roofline(q["n"], q["nnz"], q["f"], q["d"], q["cfg"], args.dtype,
                                                           q["elapsed"] / 3 * 1e3)}}
        # ... and on a STANDARD skewed generator nobody here tuned: R-MAT with the Graph500 parameters
        q = run_workload(args, "rmat", rank, world, dev, 3, 2)
        structured["rmat"] = {
            "graph": "R-MAT, Graph500 parameters a, b, c = 0.57, 0.19, 0.19 over ceil(log2 N) levels restricted to N ids, "
                     "same number of undirected pairs before coalescing, node ids randomly permuted "
                     "(synth.synthetic_graph_rmat)",
            "nnz": q["nnz"], "value": q["n"] * 3 / q["elapsed"], "unit": "nodes/s", "steps": 3,
            "ms_per_step": round(q["elapsed"] / 3 * 1e3, 3), "graph_view": q["view"],
            "roofline": {**q["roof"], "step": step_roofline(q["n"], q["nnz"], q["f"], q["d"], q["cfg"], args.dtype,
                                                           q["elapsed"] / 3 * 1e3)}}
        if args.workload == "ogbn-products" and args.dtype == "bf16":
            # ... and the headline shape in the REFERENCE's own arithmetic: fp32 storage (large/ours.py has no autocast), the
            # parity mode of every test (logits within 1e-4 of the fp64 oracle)
            a32 = argparse.Namespace(**{**vars(args), "dtype": "f32"})
            q = run_workload(a32, "uniform", rank, world, dev, 3, 1)
            ms32 = q["elapsed"] / 3 * 1e3
            structured["fp32"] = {
                "what": "the headline step with fp32 storage (the reference's arithmetic; BASELINE config 3 itself is bf16)",
                "nnz": q["nnz"], "value": q["n"] * 3 / q["elapsed"], "unit": "nodes/s", "steps": 3, "dtype": "f32",
                "ms_per_step": round(ms32, 3), "loss": q["loss"], "graph_view": q["view"],
                "roofline": {**q["roof"], "step": step_roofline(q["n"], q["nnz"], q["f"], q["d"], q["cfg"], "f32", ms32)}}
        if args.workload == "ogbn-products" and not args.nodes and not args.no_minibatch_leg:
            # ... and the reference's OTHER way through the same model (large/main-batch.py, the recipe of large/run.sh:15-19):
            # one epoch of random-partition mini-batches, the trainer's loop lines verbatim (= `--mode minibatch`); last, because
            # it sets the host thread count the launcher uses for that trainer
            # (TWO untimed epochs: a batch size is captured the second time it is seen — the ragged last batch of an epoch in
            # epoch 2 — and a capture, ~0.1 s, is a one-off like the CSR build; r05 ran one, and the driver's line carried it)
            q = run_minibatch(args, dev, 3, 2)
            structured["minibatch_epoch"] = {
                "what": "one EPOCH of large/main-batch.py:129-151 on the same uniform graph: random partitions of "
                        f"{args.batch_size} nodes, induced subgraph + its CSR per batch (sgf_subgraph_csr_*), model steps replayed "
                        "as hipGraphs (sgformer_amd/graphed.py), the trainer's own host lines, loss lines and Adam",
                "value": q["n"] * 3 / q["elapsed"], "unit": "nodes/s", "steps": 3, "ms_per_epoch": round(q["elapsed"] / 3 * 1e3, 3),
                "loss": q["loss"], "minibatch": q["breakdown"], "roofline": q["roof"]}

    if rank == 0:
        n, f, c, d, weak = r["n"], r["f"], r["c"], r["d"], r["weak"]
        ms = r["elapsed"] / args.steps * 1e3
        gname = {"uniform": "uniform random graph", "community": "community graph with shuffled node ids",
                 "powerlaw": "power-law community graph with global hubs and shuffled node ids",
                 "rmat": "R-MAT graph (Graph500 a, b, c = 0.57, 0.19, 0.19) with shuffled node ids"}[args.graph]
        step_rf = step_roofline(n, r["nnz"] * (world if weak else 1), f, d, r["cfg"], args.dtype, ms)
        roof = {**r["roof"], "step": step_rf} if r["roof"] is not None else {"step": step_rf}
        line = {
            "metric": f"SGFormer fwd+bwd nodes/sec on {args.workload} full-graph",
            "value": n * args.steps / r["elapsed"], "unit": "nodes/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True,
            "scaling": "weak" if weak else "strong", "vs_baseline": None, "dtype": args.dtype,
            "data": "synthetic",
            "config": {"workload": f"{args.workload}-shaped {gname}, full-graph "
                                   f"training step (fwd + log_softmax/NLL on the training rows + bwd + Adam), "
                                   f"{'100M' if 'papers' in args.workload else 'large'}/run.sh recipe, dropout "
                                   + ("0" if not any(r["dropout"]) else f"{r['dropout'][0]} (attention branch) / "
                                      f"{r['dropout'][1]} (GCN branch) as in the recipe, sgf_dropout")
                                   + (f"; {n // world:,} nodes per rank, rows generated per rank" if weak else ""),
                       "loss": {"trainer": "the trainer's own lines (large/main.py:139-141: log_softmax, row indexing, "
                                           "nn.NLLLoss) as they run under sgformer_amd.launch (launch.patch_nll_loss: F.log_softmax returns a lazy "
                                           "tensor, indexing + NLLLoss run as one pass over the training rows)",
                                "fused": "sgformer_amd.loss.log_softmax_nll (same arithmetic, one pass)",
                                "aten": "F.log_softmax + F.nll_loss on ATen's kernels"}[r["loss_mode"]],
                       "ms_per_step_with_aten_loss": None if r["ms_aten"] is None else round(r["ms_aten"], 3),
                       "ms_per_step_with_fused_loss": None if r["ms_fused"] is None else round(r["ms_fused"], 3),
                       "features_in": "fp32 features handed to the model every step (as large/main.py:130 does); the module "
                                      "caches its storage-dtype copy of a feature tensor it has already seen",
                       "roofline_target_reading": "frac is on ALGORITHMIC bytes (SURVEY.md §8d B_spmm); gather_GBps / traffic "
                                                  "give the measured-bytes reading SURVEY.md §7 allows for graphs without "
                                                  "locality: the 0.60 target is met on measured bytes (uniform graph), not on "
                                                  "algorithmic bytes",
                       "nodes": n, ("nnz_per_rank" if weak else "nnz"): r["nnz"], "features": f, "hidden": d, "classes": c,
                       "parallelism": f"node-shard x{world}" if world > 1 else "single GPU",
                       "graph_view": r["view"],
                       "prepare_graph_s": None if r["prepare_s"] is None else round(r["prepare_s"], 3),
                       "exchanged": r["exchanged"],
                       "debug_override": bool(args.nodes), **({"dry_run": True} if DRYRUN else {}),
                       **({"shared_gpu": "validation only: all ranks on cuda:0, gloo transport staged through the host"}
                          if SHARE_GPU else {})},
            "loss": r["loss"],
            "peak_mem_GB": r["peak_mem"],
            "roofline": roof,
            "legs": legs_summary(ms, roof, structured),
            "structured": structured,
            "cpu_baseline": cpu,
        }
        if cpu is not None:
            cpu["what"] = ("oracle port (oracle/sgformer_oracle.py on torch's CPU kernels), pinned to the reference at 1e-12 "
                           "(tests/test_oracle.py against the live /root/reference); the reference itself is not on this box")
        if world > 1:
            line["scaling_model"] = scaling_model(args.workload, args.dtype, world, ms)
            line["per_rank_memory_model"] = per_rank_memory_model(args.workload, world, args.dtype)
        if cpu is not None:
            line["speedup_vs_cpu_baseline"] = round(line["value"] / cpu["value"], 1)
        print(json.dumps(line), flush=True)
    if _sharded(world):
        dist.barrier()                       # every rank is done with its collectives before any of them tears the group down
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
