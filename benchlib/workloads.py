"""bench.py: the measured workloads — one full-graph training step (any BASELINE configuration, any graph generator) and one
mini-batch epoch of large/main-batch.py."""
from __future__ import annotations

import os
import sys
import time

import torch
import torch.distributed as dist
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from sgformer_amd import ops, synth  # noqa: E402
from sgformer_amd.loss import log_softmax_nll  # noqa: E402
from sgformer_amd.dist import ShardContext, shard_model, sharded_nll_loss  # noqa: E402
from sgformer_amd.ours import SGFormer  # noqa: E402

from .timers import SpmmTimer, pmc_traffic  # noqa: E402

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
        iso = None
        if getattr(model, "overlap_branches", False) and ctx is None and not medium and timer.pairs:
            # The timed steps run the attention branch on a side stream NEXT to the SpMM launches, so the events above time
            # the SpMM sharing the chip.  One more, untimed step on ONE stream gives the kernel's own launch time as well.
            in_situ = (timer.pairs, timer.bytes_alg, timer.bytes_gather, timer.kernels)
            timer.reset()
            timer.active, model.overlap_branches = True, False
            step()
            fence()
            timer.active, model.overlap_branches = False, True
            iso = timer.summary()
            timer.pairs, timer.bytes_alg, timer.bytes_gather, timer.kernels = in_situ
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
    if roof is not None and iso is not None:
        roof["single_stream"] = {"mean_launch_ms": iso["mean_launch_ms"], "achieved": iso["achieved"], "frac": iso["frac"],
                                 "gather_GBps": iso["gather_GBps"], "launches": iso["launches"],
                                 "what": "the same kernel in one extra, untimed step with both branches on ONE stream; the figures "
                                         "above are from the timed steps, where the attention branch shares the chip"}
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
                     "halo_exchanges_overlapped_per_step": getattr(ctx, "overlapped_exchanges", 0) // max(warmup + steps, 1),
                     # which way the SpMM operand travelled: the halo plan (cut-edge rows only, all_to_all) when the largest
                     # halo / remote-rows ratio over the ranks stays under SGF_HALO_MAX, else every remote row (all-gather)
                     "spmm_exchange": None if getattr(ctx, "last_halo", None) is None else
                     {"path": "halo" if ctx.last_halo[1] else "all_gather", "halo_fraction_max": round(ctx.last_halo[0], 4),
                      "halo_max": ctx.halo_max, "overlap_own_columns": bool(getattr(ctx, "overlap", False))}}
    if medium:
        cfg = dict(gnn_num_layers=4, trans_num_layers=1)     # medium/run.sh:2-7 (for the whole-step byte formula)
    out = dict(n=n, f=f, c=c, d=d, weak=weak, cfg=cfg, nnz=int(ei.shape[1]), elapsed=elapsed, loss=loss_val, ms_aten=ms_aten, ms_fused=ms_fused, loss_mode=state["mode"],
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
    from sgformer_amd import staging
    x = staging.resident(x.to(dev))                                 # launch.patch_resident_features: features resident on the GPU,
    true_label = staging.staged(y).unsqueeze(1)                     # labels a host tensor whose picked rows travel on the prep stream
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
