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
                  from the rocprofv3 PMC passes of THIS kernel on THIS workload (profiles/r05_spmm_pmc.json,
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

# HBM bytes per SpMM launch from rocprofv3 PMC passes (separate --pmc FETCH_SIZE / WRITE_SIZE runs of
# scripts/spmm_pmc_target.py; FETCH_SIZE x2 per the gfx950 correction of MI355X_MICROARCH.md §HBM).  PMC
# counters cannot be collected from inside this process, so the figures live in a tracked file written
# from those passes, keyed on graph kind / dtype / kernel.
PMC_FILE = os.path.join(ROOT, "profiles", "r05_spmm_pmc.json")
SPMM_SOURCES = ("spmm.hip", "spmm_tile.hip", "spmm_pack.hip", "spmm_plan.hip", "spmm_shared.h")


def spmm_source_sha16() -> str:
    """sha256 (first 16 hex digits) of the SpMM sources the PMC passes were taken with."""
    import hashlib
    h = hashlib.sha256()
    for name in SPMM_SOURCES:
        with open(os.path.join(ROOT, "sgformer_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def pmc_traffic(graph_kind: str, dtype: str, kernel: str, reordered: bool):
    """(HBM bytes per launch, source) — or (None, why) when no pass exists or the passes are STALE: the file records
    the hash of the kernel sources it was measured with, and a number measured on other code is not reported."""
    try:
        table = json.load(open(PMC_FILE))
    except (OSError, ValueError):
        return None, None
    if table.get("_source_sha16") != spmm_source_sha16():
        return None, "profiles/r05_spmm_pmc.json is older than csrc/spmm*.hip: re-run scripts/pmc_passes.sh"
    tag = "reordered" if reordered else "given"
    e = table.get(f"{graph_kind}/{dtype}/{kernel}/{tag}")
    if not e:
        return None, None
    total = e["hbm_bytes_per_launch"]
    if kernel in ("k_spmm_row", "k_spmm_seg_bf16x2"):
        # one sgf_spmm call = the row kernel + the long-row path (hub rows: k_spmm_long_seg / _fin), timed together by the
        # HIP events above — so their traffic is reported together too (R-MAT: 29.8 + 24.1 + 0.1 GB)
        for extra in ("k_spmm_long_seg", "k_spmm_long_fin"):
            x = table.get(f"{graph_kind}/{dtype}/{extra}/{tag}") or table.get(f"{graph_kind}/{dtype}/{extra}/given")
            if x:
                total += x["hbm_bytes_per_launch"]
    return total, "profiles/r05_spmm_pmc.json"


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
    if workload == "cora":
        return _cpu_baseline_cora(seed)
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


def _cpu_baseline_cora(seed: int):
    """BASELINE config 1 on the host: the oracle's restatement of medium/ours.py + medium/models.py GCN (oracle.medium_forward),
    Cora shape at its full size (2 708 nodes), medium/run.sh:2-7 recipe, fp32, dropout 0, fwd + loss + bwd."""
    from oracle import sgformer_oracle as O
    from sgformer_amd import ours_medium as M
    n, avg_deg, f, c, d = synth.SHAPES["cora"]
    cfg = dict(num_layers=1, alpha=0.5, use_bn=False, use_residual=False, use_weight=False, graph_weight=0.8)
    torch.manual_seed(seed)
    gnn = M.GCN(f, d, d, num_layers=4, dropout=0.0, use_bn=False)
    m = M.SGFormer(f, d, c, dropout=0.0, gnn=gnn, **cfg)
    p = {k: v.detach().clone().requires_grad_(v.is_floating_point() and "running" not in k) for k, v in m.state_dict().items()}
    ei = synth.synthetic_graph(n, avg_deg, seed=seed)[:, :-n]
    x = (torch.rand(n, f, generator=torch.Generator().manual_seed(seed)) < 0.0127).float()
    _, y, idx = synth.synthetic_task(n, 4, c, seed=seed)
    best = None
    for threads in (1, 4, 8, 16):
        if threads > (os.cpu_count() or 1):
            break
        torch.set_num_threads(threads)
        times = []
        for _ in range(6):
            for v in p.values():
                v.grad = None
            t0 = time.perf_counter()
            O.nll_loss(O.medium_forward(p, x, ei, cfg, training=True), y, idx).backward()
            times.append(time.perf_counter() - t0)
        dt = sorted(times[1:])[len(times[1:]) // 2]
        if best is None or dt < best[0]:
            best = (dt, threads)
    dt, threads = best
    return {"value": n / dt, "unit": "nodes/s", "cores": threads, "kind": "port",
            "sample": f"Cora-shaped graph at its full size (N = {n}, nnz = {int(ei.shape[1])}), medium/run.sh:2-7 recipe, fp32, "
                      f"dropout 0, fwd+loss+bwd, median of 5 after 1 warm-up, {dt * 1e3:.1f} ms/step, {threads} host threads "
                      f"(fastest of 1 / 4 / 8 / 16)"}


class SpmmTimer:
    """HIP-event timing of every SpMM launch on the launch stream (torch's current stream)."""

    def __init__(self):
        self.pairs, self.bytes_alg, self.bytes_gather, self.active, self.kernels = [], [], [], False, []
        self._orig = (ops.K.spmm, getattr(ops.K, "spmm_blocked", None), getattr(ops.K, "spmm_tile", None))

    def _wrap(self, orig, blocked):
        timer = self

        def timed(rowptr, b, *rest, **kw):
            # K.spmm(rowptr, colind, val, x, n_rows, ...) / K.spmm_blocked(rowptr, plan, x, n_rows, ...)
            if not timer.active or torch.cuda.is_current_stream_capturing():
                return orig(rowptr, b, *rest, **kw)
            x, n_rows = (rest[0], rest[1]) if blocked else (rest[1], rest[2])
            nnz = int(b.nnz) if blocked else b.numel()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            y = orig(rowptr, b, *rest, **kw)
            e1.record()
            s, d = x.element_size(), x.shape[1]
            bf16 = x.dtype == torch.bfloat16
            # the library's dispatch (csrc/spmm.hip::launch): stream kernel for re-ordered graphs and for bf16 rows of
            # 65-128 elements, wave per row above 128, sub-wave per row below
            timer.kernels.append("k_spmm_blk" if blocked else (
                "k_spmm_seg_bf16x2" if bf16 and d % 8 == 0 and (kw.get("stream_hint") and d > 128 or 64 < d <= 128)
                else ("k_spmm_sub" if d <= 128 else "k_spmm_row")))
            timer.pairs.append((e0, e1))
            timer.bytes_alg.append(nnz * 8 + (n_rows + 1) * 8 + x.shape[0] * d * s + n_rows * d * s)
            timer.bytes_gather.append(nnz * (8 + d * s) + (n_rows + 1) * 8 + n_rows * d * s)
            return y

        return timed

    def _wrap_tile(self, orig):
        timer = self

        def timed(plan, x, n_rows, *rest, **kw):
            if not timer.active or torch.cuda.is_current_stream_capturing():
                return orig(plan, x, n_rows, *rest, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            y = orig(plan, x, n_rows, *rest, **kw)
            e1.record()
            s, d, nnz = x.element_size(), x.shape[1], int(plan.nnz)
            timer.kernels.append("k_spmm_tile_bf16")
            timer.pairs.append((e0, e1))
            timer.bytes_alg.append(nnz * 8 + (n_rows + 1) * 8 + x.shape[0] * d * s + n_rows * d * s)
            timer.bytes_gather.append(nnz * (8 + d * s) + (n_rows + 1) * 8 + n_rows * d * s)
            return y

        return timed

    def install(self):
        ops.K.spmm = self._wrap(self._orig[0], False)
        ops.K.spmm_blocked = self._wrap(self._orig[1], True)
        ops.K.spmm_tile = self._wrap_tile(self._orig[2])

    def uninstall(self):
        ops.K.spmm, ops.K.spmm_blocked, ops.K.spmm_tile = self._orig

    def reset(self):
        self.pairs, self.bytes_alg, self.bytes_gather, self.kernels = [], [], [], []

    def summary(self):
        if not self.pairs:
            return None
        ms = [a.elapsed_time(b) for a, b in self.pairs]
        mean_ms = sum(ms) / len(ms)
        alg = sum(self.bytes_alg) / len(self.bytes_alg)
        gat = sum(self.bytes_gather) / len(self.bytes_gather)
        achieved = alg / (mean_ms * 1e-3) / 1e9
        kern = max(set(self.kernels), key=self.kernels.count)
        entry = {"k_spmm_blk": "sgf_spmm_blocked", "k_spmm_seg_bf16x2": "sgf_spmm_stream",
                 "k_spmm_tile_bf16": "sgf_spmm_tile"}.get(kern, "sgf_spmm")
        return {"kernel": f"{kern} ({entry})", "bound": "hbm", "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": None, "launches": len(ms), "mean_launch_ms": round(mean_ms, 4),
                "algorithmic_bytes": int(alg), "gather_bytes": int(gat),
                "gather_GBps": round(gat / (mean_ms * 1e-3) / 1e9, 1)}


def _sharded(world: int) -> bool:
    """Node-sharded path: always for N > 1; for N = 1 only when a test asks for it (SGF_BENCH_FORCE_SHARD=1 under
    torch.distributed.run: the RCCL init, the ShardContext and every collective of the step with a single rank)."""
    return world > 1 or (os.environ.get("SGF_BENCH_FORCE_SHARD") == "1" and "MASTER_PORT" in os.environ)


def make_inputs(workload: str, nodes: int, seed: int, rank: int, world: int, dev, graph: str = "uniform"):
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
        if _sharded(world):
            ctx = ShardContext(n, local_edges=True)
    else:
        gen = {"community": synth.synthetic_graph_community, "powerlaw": synth.synthetic_graph_community_powerlaw,
               "uniform": synth.synthetic_graph, "rmat": synth.synthetic_graph_rmat}[graph]
        ei = gen(n, avg_deg, seed=seed, device=dev)
        x, y, train_idx = synth.synthetic_task(n, f, c, seed=seed)
        n_train = train_idx.numel()
        if _sharded(world):
            ctx = ShardContext(n)
            x, y, train_idx = ctx.shard_rows(x), ctx.shard_rows(y), ctx.local_index(train_idx)
    return n, f, c, d, cfg, weak, ei, x, y, train_idx, n_train, ctx


def run_workload(args, graph_kind, rank, world, dev, steps, warmup, with_aten=False):
    """Build the synthetic inputs + model for one graph kind, run `warmup` untimed and `steps` timed training
    steps (barrier + synchronize on both sides), return the measurements."""
    n, f, c, d, cfg, weak, ei, x, y, train_idx, n_train, ctx = make_inputs(args.workload, args.nodes, args.seed,
                                                                           rank, world, dev, graph_kind)
    medium = args.workload == "cora"
    if medium and (args.dtype != "f32" or ctx is not None):
        raise SystemExit("--workload cora is BASELINE config 1 (medium/ours.py): fp32, one GPU")
    dtype = torch.float32 if args.dtype == "f32" else torch.bfloat16
    # the features are handed to the model in fp32 EVERY step, exactly as an unchanged trainer does (large/main.py:130:
    # model(dataset.graph['node_feat'], ...)); the module keeps its storage-dtype (and row-permuted / zero-padded) copy of a
    # feature tensor it has seen before (SGFormer.forward: keyed on the tensor's identity and version) — the features of a
    # full-graph run are constant data, like the CSR
    x, y, train_idx = x.to(dev), y.to(dev), train_idx.to(dev)

    torch.manual_seed(args.seed)
    p_trans, p_gnn = synth.RECIPE_DROPOUT.get(args.workload, (0.0, 0.0)) if args.dropout == "recipe" else (0.0, 0.0)
    # bf16 = bf16 activation storage with fp32 master weights and fp32 accumulation everywhere
    if medium:
        # BASELINE config 1: medium/ours.py SGFormer with the GCN backbone, medium/run.sh:2-7 (1 attention layer without
        # LayerNorm / residual / Wv, GCN num_layers 4 hidden 64 without BatchNorm, graph_weight 0.8, alpha 0.5); the trainer
        # symmetrises the edge list and adds no self-loops (medium/main.py:94) — GCNConv adds them itself; bag-of-words
        # features (--no_feat_norm).  model(data) reads data.graph[...] (medium/ours.py:134-136).
        from sgformer_amd import ours_medium as M
        ei = ei[:, :-n].contiguous()
        gx = torch.Generator().manual_seed(args.seed)
        x = (torch.rand(n, f, generator=gx) < 0.0127).float().to(dev)
        gnn = M.GCN(f, d, d, num_layers=4, dropout=p_gnn, use_bn=False)
        core = M.SGFormer(f, d, c, num_layers=1, alpha=0.5, dropout=p_trans, use_bn=False, use_residual=False, use_weight=False,
                          use_graph=True, graph_weight=0.8, gnn=gnn).to(dev)

        class _Data:
            def __init__(self, feat, edges):
                self.graph = {"node_feat": feat, "edge_index": edges, "num_nodes": feat.shape[0]}

        class _Wrap(torch.nn.Module):       # bench's step calls model(x, edge_index); the medium module takes the Data object
            def __init__(self, core_):
                super().__init__()
                self.core, self.params1, self.params2 = core_, core_.params1, core_.params2

            def forward(self, feat, edges):
                return self.core(_Data(feat, edges))
        model = _Wrap(core)
    else:
        model = SGFormer(f, d, c, trans_dropout=p_trans, gnn_dropout=p_gnn,
                         compute_dtype=None if args.dtype == "f32" else dtype, **cfg).to(dev)
        model.logits_dtype = torch.float32     # (fp32 features in -> fp32 logits out, as under sgformer_amd.launch)
    if ctx is not None:
        shard_model(model, ctx)
    # the optimizer exactly as the trainer constructs it (large/main.py:114-119); under sgformer_amd.launch — and here —
    # torch's single-kernel form of the same arithmetic is the default for CUDA parameters (launch.patch_adam)
    from sgformer_amd import launch as _launch_adam
    _launch_adam.patch_adam()
    opt = torch.optim.Adam([{"params": model.params1, "weight_decay": 1e-5},
                            {"params": model.params2, "weight_decay": 1e-5}], lr=0.01)
    model.train()
    # per-graph, not per-step: CSR, node order, row-block plan (the trainers get the same lazily in their
    # first two epochs) — outside the timed region like every other one-off
    view_stats, t_prep = None, None
    if ctx is None and not medium:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        view_stats = dict(ops.prepare_graph(ei, n).stats)
        torch.cuda.synchronize()
        t_prep = time.perf_counter() - t0

    from sgformer_amd import launch as _launch
    state = {"mode": "aten" if args.aten_loss else args.loss}

    def step():
        opt.zero_grad(set_to_none=True)
        logits = model(x, ei)
        if ctx is not None:
            loss = sharded_nll_loss(logits, y, train_idx, n_train)
        elif state["mode"] == "fused":   # the same arithmetic in one pass (sgf_nll_fwd / sgf_nll_bwd)
            loss = log_softmax_nll(logits, y, train_idx)
        else:   # the three lines of large/main.py:139-141 as the trainer writes them ('trainer': F.nll_loss is the
            # launcher's gather form while the step runs, 'aten': ATen's kernels)
            loss = F.nll_loss(F.log_softmax(logits.float(), dim=1)[train_idx], y[train_idx])
        loss.backward()
        if ctx is not None:
            ctx.sync_grads(model.parameters())
        opt.step()
        return loss

    timer = SpmmTimer()
    timer.install()
    if state["mode"] == "trainer" and ctx is None:
        _launch.patch_nll_loss()
    try:
        for _ in range(warmup):
            step()

        def fence():
            torch.cuda.synchronize()
            if _sharded(world):
                dist.barrier()
            torch.cuda.synchronize()

        fence()
        timer.active = True
        t0 = time.perf_counter()
        for _ in range(steps):
            loss = step()
        fence()
        elapsed = time.perf_counter() - t0
        timer.active = False   # (the extra ATen-loss steps below are not part of the roofline sample)
        loss_val = float(loss.detach())
        if _sharded(world):
            t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t)
            lt = torch.tensor([loss_val], device=dev, dtype=torch.float64)
            dist.all_reduce(lt)
            loss_val = float(lt)
        ms_aten = ms_fused = None
        if with_aten and world == 1:
            # transparency: the same step with the other two loss forms
            headline = state["mode"]
            _launch.unpatch_nll_loss()

            def extra(mode):
                state["mode"] = mode
                step()
                fence()
                t1 = time.perf_counter()
                for _ in range(min(steps, 5)):
                    step()
                fence()
                return (time.perf_counter() - t1) / min(steps, 5) * 1e3

            if headline != "aten":
                ms_aten = extra("aten")
            if headline != "fused":
                ms_fused = extra("fused")
            state["mode"] = headline
    finally:
        _launch.unpatch_nll_loss()
        timer.uninstall()
    roof = timer.summary()
    if roof is not None and world == 1 and not args.nodes:
        kern = roof["kernel"].split(" ")[0]
        roof["traffic"], src = pmc_traffic(f"{args.workload}:{graph_kind}", args.dtype, kern,
                                           bool(view_stats and view_stats.get("reordered")))
        if src:
            roof["traffic_source"] = src
    exchanged = None
    if ctx is not None:
        exchanged = {"halo_bytes_sent_per_step": ctx.bytes_halo_sent // max(warmup + steps, 1),
                     "all_gather_bytes_per_step": ctx.bytes_all_gathered // max(warmup + steps, 1),
                     "all_reduce_bytes_per_step": ctx.bytes_all_reduced // max(warmup + steps, 1),
                     "repartition_bytes_per_step": ctx.bytes_repartition // max(warmup + steps, 1),
                     # exchanges that actually ran split (own-column product while the halo rows travelled), per step —
                     # 0 when the halo plan is off (all-gather fallback) or this rank has no halo
                     "halo_exchanges_overlapped_per_step": getattr(ctx, "overlapped_exchanges", 0) // max(warmup + steps, 1)}
    out = dict(n=n, f=f, c=c, d=d, weak=weak, nnz=int(ei.shape[1]), elapsed=elapsed, loss=loss_val, ms_aten=ms_aten, ms_fused=ms_fused, loss_mode=state["mode"],
               dropout=(p_trans, p_gnn),
               roof=roof, view=view_stats, prepare_s=t_prep, exchanged=exchanged,
               peak_mem=round(torch.cuda.max_memory_allocated() / 2 ** 30, 2))
    del model, opt, x, y
    ops.graph_cache.clear()
    torch.cuda.empty_cache()
    return out


def run_minibatch(args, dev, steps, warmup):
    """One step = one EPOCH of large/main-batch.py:129-151 as the unchanged trainer runs it under sgformer_amd.launch
    (features resident on the GPU: launch.patch_resident_features; per-batch induced subgraph on the GPU: batching.subgraph
    = torch_geometric.utils.subgraph's semantics; 16 host threads; labels and masks on the HOST as the trainer keeps them):
        idx = randperm(n);  per batch:  train_mask[idx_i], x[idx_i].to(device), subgraph(idx_i, edge_index, relabel), y[idx_i]
        .to(device), model(x_i, edge_index_i), log_softmax, criterion(out_i[train_mask_i], y_i[train_mask_i]), backward, step
    — the trainer's lines verbatim, including its boolean-mask indexing (one device->host read per batch).  The per-batch
    breakdown comes from HIP events on the launch stream (GPU timeline, gaps included) and host timers (time to ISSUE)."""
    from sgformer_amd import batching, launch
    launch.limit_host_threads()
    n, avg_deg, f, c, d = synth.SHAPES[args.workload]
    if args.nodes:
        n = args.nodes
    cfg = dict(synth.RECIPES.get(args.workload, synth.RECIPES["ogbn-products"]))
    gen = {"community": synth.synthetic_graph_community, "powerlaw": synth.synthetic_graph_community_powerlaw,
           "uniform": synth.synthetic_graph, "rmat": synth.synthetic_graph_rmat}[args.graph]
    ei = gen(n, avg_deg, seed=args.seed, device=dev).cpu()          # the dataset lives on the HOST (main-batch.py:43-99)
    x, y, train_idx = synth.synthetic_task(n, f, c, seed=args.seed)
    x = x.to(dev)                                                   # launch.patch_resident_features
    true_label = y.unsqueeze(1)
    train_mask = torch.zeros(n, dtype=torch.bool)
    train_mask[train_idx] = True
    dtype = None if args.dtype == "f32" else torch.bfloat16
    torch.manual_seed(args.seed)
    model = SGFormer(f, d, c, trans_dropout=0.0, gnn_dropout=0.0, compute_dtype=dtype, **cfg).to(dev)
    launch.patch_adam()
    opt = torch.optim.Adam(model.parameters(), weight_decay=1e-5, lr=0.01)      # main-batch.py:125-127 (one group)
    criterion = torch.nn.NLLLoss()
    bs = args.batch_size
    num_batch = n // bs + (n % bs > 0)
    marks = ("gather", "subgraph", "forward", "loss_backward", "optimizer")
    ev, host = [], {k: 0.0 for k in marks}
    gen_cpu = torch.Generator().manual_seed(args.seed)

    def epoch(record):
        model.train()
        idx = torch.randperm(n, generator=gen_cpu)
        for i in range(num_batch):
            e = [torch.cuda.Event(enable_timing=True) for _ in range(6)] if record else None
            t = [time.perf_counter()]
            if record:
                e[0].record()
            idx_i = idx[i * bs:(i + 1) * bs]
            train_mask_i = train_mask[idx_i]
            x_i = x[idx_i].to(dev)
            y_i = true_label[idx_i].to(dev)
            t.append(time.perf_counter())
            if record:
                e[1].record()
            ei_i, _ = batching.subgraph(idx_i, ei, num_nodes=n, relabel_nodes=True)
            ei_i = ei_i.to(dev)
            t.append(time.perf_counter())
            if record:
                e[2].record()
            opt.zero_grad()
            out_i = model(x_i, ei_i)
            t.append(time.perf_counter())
            if record:
                e[3].record()
            out_i = F.log_softmax(out_i, dim=1)
            loss = criterion(out_i[train_mask_i], y_i.squeeze(1)[train_mask_i])
            loss.backward()
            t.append(time.perf_counter())
            if record:
                e[4].record()
            opt.step()
            t.append(time.perf_counter())
            if record:
                e[5].record()
                ev.append(e)
                for k, a, b in zip(marks, t, t[1:]):
                    host[k] += b - a
        return loss

    launch.patch_nll_loss()
    timer = SpmmTimer()
    timer.install()
    try:
        for _ in range(warmup):
            epoch(False)
        torch.cuda.synchronize()
        from sgformer_amd import graphed
        replayed, replays0 = graphed.enabled(), graphed.counters["replays"]
        timer.active = not replayed
        t0 = time.perf_counter()
        for _ in range(steps):
            loss = epoch(True)
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        timer.active = False
        replays = graphed.counters["replays"] - replays0          # (0: enabled, but no batch was eligible / the capture failed)
        if replayed:
            # the timed epochs replay captured steps (sgformer_amd/graphed.py): no per-launch events in there.  The SpMM
            # launches are timed in one more, UNTIMED epoch of eager steps — the same kernels on the same batches' sizes.
            keep, n_ev = os.environ.get("SGF_BATCH_GRAPH"), len(ev)
            os.environ["SGF_BATCH_GRAPH"] = "0"
            try:
                timer.active = True
                epoch(False)
                torch.cuda.synchronize()
                timer.active = False
            finally:
                if keep is None:
                    del os.environ["SGF_BATCH_GRAPH"]
                else:
                    os.environ["SGF_BATCH_GRAPH"] = keep
            del ev[n_ev:]
    finally:
        launch.unpatch_nll_loss()
        timer.uninstall()
    gpu = {k: 0.0 for k in marks}
    for e in ev:
        for k, a, b in zip(marks, e, e[1:]):
            gpu[k] += a.elapsed_time(b)
    nb = max(len(ev), 1)
    breakdown = {"batches_per_epoch": num_batch, "batch_nodes": bs, "steps_replayed_as_hip_graphs": bool(replayed and replays > 0),
                 "replayed_steps_in_the_timed_epochs": int(replays),
                 "per_batch_ms_on_the_gpu_timeline": {k: round(v / nb, 3) for k, v in gpu.items()},
                 "per_batch_ms_host_issue": {k: round(v / nb * 1e3, 3) for k, v in host.items()},
                 "per_batch_ms_wall": round(elapsed / nb * 1e3, 3)}
    out = dict(n=n, f=f, c=c, d=d, nnz=int(ei.shape[1]), elapsed=elapsed, loss=float(loss.detach()), roof=timer.summary(),
               breakdown=breakdown, peak_mem=round(torch.cuda.max_memory_allocated() / 2 ** 30, 2))
    del model, opt, x
    ops.graph_cache.clear()
    torch.cuda.empty_cache()
    return out


# single-GPU step times measured on MI355X (ms; profiles/r05_bench_*.json) that the scaling MODEL below starts from
MEASURED_1GPU_MS = {("pokec", "bf16"): 30.0, ("pokec", "f32"): 95.0, ("ogbn-products", "bf16"): 91.5,
                    ("papers100M-weak", "bf16"): 232.0, ("papers100M-shard8", "bf16"): 232.0, ("ogbn-arxiv", "f32"): 9.9}
XGMI_LINK_GBS = 153.0      # per direction and link, 7 links per GPU (MI355X_MICROARCH.md); 0.8 of it assumed reachable


def scaling_model(workload: str, dtype: str, world: int):
    """A MODEL of the node-sharded step on `world` GPUs of one node — bytes over xGMI links, NOT a measurement (this build
    never had more than one GPU; SCALE_r0x.json holds the driver's real numbers when an 8-GPU node was available).
    Exchanges per step (sgformer_amd/dist.py): per SpMM launch (2 per GCN layer: forward and backward) every rank receives
    the other ranks' rows of X — all-gather of N d s bytes on a graph without locality (uniform generator; the halo plan
    sends only the cut-edge rows on a graph sgf_reorder can partition); per attention pass one all-reduce of d^2 + O(d)
    floats, per BatchNorm 2 d + 1 floats each way, the parameter gradients once.  xGMI is point to point: a rank's P - 1
    incoming shards arrive on P - 1 different links in parallel, so an all-gather costs one shard over one link."""
    n, _, f, c, d = synth.SHAPES[workload]
    weak = workload.endswith("-weak")
    cfg = synth.RECIPES.get(workload, synth.RECIPES["ogbn-products"])
    s = 4 if dtype == "f32" else 2
    n_total = n * world if weak else n
    shard_rows = n_total // world
    lg = cfg["gnn_num_layers"]
    spmm_launches = 2 * lg
    link = XGMI_LINK_GBS * 0.8 * 1e9
    shard_bytes = shard_rows * d * s
    t_all_gather = spmm_launches * shard_bytes / link if world > 1 else 0.0
    small = (2 * (d * d + 2 * d + 2) + (lg + 1) * 2 * (2 * d + 1)) * 4           # attention fwd + bwd, BatchNorm fwd + bwd
    n_params = 2 * f * d + 3 * d * d + lg * (2 * d * d if cfg.get("gnn_use_init") else d * d) + d * c
    t_small = (2 + 2 * (lg + 1)) * 30e-6 + (small + n_params * 4) / link if world > 1 else 0.0   # ~30 us per tiny collective
    base = MEASURED_1GPU_MS.get((workload, dtype))
    out = {"label": "MODEL — not a measurement: no multi-GPU hardware was available to this build",
           "per_rank_rows": shard_rows, "spmm_launches_per_step": spmm_launches,
           "all_gather_bytes_received_per_rank_per_step": int(spmm_launches * shard_bytes * (world - 1)),
           "all_reduce_bytes_per_step": int(small + n_params * 4),
           "xgmi_link_GBps_assumed": round(XGMI_LINK_GBS * 0.8, 1),
           "t_all_gather_ms_no_overlap": round(t_all_gather * 1e3, 3), "t_small_collectives_ms": round(t_small * 1e3, 3)}
    if base is not None:
        compute = base if weak else base / world
        step = compute + (t_all_gather + t_small) * 1e3
        out.update({"measured_1gpu_ms": base, "compute_ms_per_rank": round(compute, 3), "modelled_step_ms": round(step, 3),
                    "modelled_nodes_per_s": round(n_total / (step * 1e-3)),
                    "modelled_efficiency": round((base / step) if weak else (base / (step * world)), 3),
                    "note": "compute share = the measured 1-GPU step (weak: unchanged; strong: / world, optimistic for the "
                            "SpMM on a uniform graph, whose halo rows do not shrink); exchanges counted WITHOUT overlap "
                            "(dist.py overlaps the own-column product with the halo exchange)"})
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
        steps, warmup = max(1, min(args.steps, 5)), max(1, min(args.warmup, 2))
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
            "prepare_graph_s": None if q["prepare_s"] is None else round(q["prepare_s"], 3), "roofline": q["roof"]}
        # the same once more with a power-law structure: community sizes 16-4096 (truncated Pareto), heavy-tailed
        # endpoints inside a community, 3 % of the pairs to global hubs (rows of tens of thousands of entries)
        q = run_workload(args, "powerlaw", rank, world, dev, 3, 1)
        structured["powerlaw"] = {
            "graph": "same N / degree; community sizes 16-4096 by a truncated Pareto law, local hubs, 3 % of the pairs "
                     "to global hubs, node ids randomly permuted (synth.synthetic_graph_community_powerlaw)",
            "nnz": q["nnz"], "value": q["n"] * 3 / q["elapsed"], "unit": "nodes/s", "steps": 3,
            "ms_per_step": round(q["elapsed"] / 3 * 1e3, 3), "graph_view": q["view"], "roofline": q["roof"]}
        # ... and on a STANDARD skewed generator nobody here tuned: R-MAT with the Graph500 parameters
        q = run_workload(args, "rmat", rank, world, dev, 3, 2)
        structured["rmat"] = {
            "graph": "R-MAT, Graph500 parameters a, b, c = 0.57, 0.19, 0.19 over ceil(log2 N) levels restricted to N ids, "
                     "same number of undirected pairs before coalescing, node ids randomly permuted "
                     "(synth.synthetic_graph_rmat)",
            "nnz": q["nnz"], "value": q["n"] * 3 / q["elapsed"], "unit": "nodes/s", "steps": 3,
            "ms_per_step": round(q["elapsed"] / 3 * 1e3, 3), "graph_view": q["view"], "roofline": q["roof"]}
        if args.workload == "ogbn-products" and not args.nodes and not args.no_minibatch_leg:
            # ... and the reference's OTHER way through the same model (large/main-batch.py, the recipe of large/run.sh:15-19):
            # one epoch of random-partition mini-batches, the trainer's loop lines verbatim (= `--mode minibatch`); last, because
            # it sets the host thread count the launcher uses for that trainer
            q = run_minibatch(args, dev, 3, 1)
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
            "roofline": r["roof"],
            "structured": structured,
            "cpu_baseline": cpu,
        }
        if world > 1:
            line["scaling_model"] = scaling_model(args.workload, args.dtype, world)
        if cpu is not None:
            line["speedup_vs_cpu_baseline"] = round(line["value"] / cpu["value"], 1)
        print(json.dumps(line), flush=True)
    if _sharded(world):
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
