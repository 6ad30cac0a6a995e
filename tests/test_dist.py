"""world_size-2 (and 3) gloo tests of the node-sharded path (sgformer_amd/dist.py) on CPU.

The compute under the collectives is the CPU kernel table of tests/cpu_kernels.py, so this checks
the exchange logic itself: equal contiguous partition with padding, all-gather indexed by global
node id, all-reduced attention / BatchNorm partials, globally normalised loss, one flat gradient
all-reduce — against the single-process oracle on the whole graph.
"""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, cfg_name, directed, ret, chunk_cols="4"):
    sys.path.insert(0, ROOT)
    # d = 16 in this test: 4-column chunks -> the pipelined 4-chunk all-gather / SpMM path runs
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), SGF_DIST_CHUNK_COLS=chunk_cols)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(2)
        from oracle import sgformer_oracle as O
        from sgformer_amd import ops
        from sgformer_amd.dist import ShardContext, shard_model, sharded_nll_loss
        from sgformer_amd.ours import SGFormer
        from tests.cpu_kernels import CpuKernels
        from tests.test_host import CONFIGS

        ops.set_kernels(CpuKernels())
        cfg = CONFIGS[cfg_name]
        n, f, d, c = 203, 10, 16, 4           # 203 is not divisible by 2 or 3: exercises padding
        torch.manual_seed(5)
        x = torch.randn(n, f)
        ei = O.synthetic_graph(n, 5.0, seed=4, directed=directed)
        y = torch.randint(0, c, (n,))
        idx = torch.randperm(n)[: n // 2]
        p = O.init_params(cfg, f, d, c, seed=6)

        ctx = ShardContext(n)
        m = SGFormer(f, d, c, trans_dropout=0.0, gnn_dropout=0.0, **cfg)
        m.load_state_dict({**m.state_dict(), **p})
        shard_model(m, ctx)
        m.train()
        logits = m(ctx.shard_rows(x), ei)
        loss = sharded_nll_loss(logits, ctx.shard_rows(y), ctx.local_index(idx), idx.numel())
        loss.backward()
        ctx.sync_grads(m.parameters())
        total = loss.detach().clone()
        dist.all_reduce(total)

        p64 = {k: v.double().requires_grad_(v.is_floating_point() and "running" not in k) for k, v in p.items()}
        stats = {}
        ref = O.sgformer_forward(p64, x.double(), ei, cfg, training=True, bn_stats=stats)
        lref = O.nll_loss(ref, y, idx)
        lref.backward()
        errs = {"logits": float((logits.detach().double() - ref.detach()[ctx.r0:ctx.r1]).abs().max()),
                "loss": abs(float(total) - float(lref))}
        gmax = max(float(v.grad.norm()) for v in p64.values() if v.grad is not None)
        gerr = 0.0
        for k, prm in m.named_parameters():
            if p64[k].grad is not None:
                e = float((prm.grad.double() - p64[k].grad).norm())
                gerr = max(gerr, e / (float(p64[k].grad.norm()) + 1e-3 * gmax))
        errs["grad"] = gerr
        sd = m.state_dict()
        rs = 0.0
        for key, (mu, vu) in stats.items():
            rs = max(rs, float((sd[key + ".running_var"].double() - (0.9 * p[key + ".running_var"].double() + 0.1 * vu)).abs().max()))
        errs["running_var"] = rs
        errs["gathered"] = ctx.bytes_all_gathered
        errs["reduced"] = ctx.bytes_all_reduced
        ret[rank] = errs
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,cfg_name,directed,chunk_cols", [(2, "products", False, "4"), (2, "heads_cat", True, "4"),
                                                                (3, "arxiv", False, "8"), (2, "products", False, "64")])
def test_sharded_step_matches_full_graph(world, cfg_name, directed, chunk_cols):
    """chunk_cols 4 / 8: the SpMM operand (d = 16) is all-gathered in 4 / 2 pipelined column chunks
    (async all-gathers, chunk SpMMs into column slices); 64: the single-gather path."""
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, cfg_name, directed, ret, chunk_cols), nprocs=world, join=True)
    assert len(ret) == world
    for rank in range(world):
        e = ret[rank]
        assert e["logits"] < 5e-5, e
        assert e["loss"] < 1e-5, e
        assert e["grad"] < 2e-3, e
        assert e["running_var"] < 1e-5, e
        assert e["gathered"] > 0 and e["reduced"] > 0


def _worker_halo(rank, world, port, directed, ret, overlap="1"):
    """A graph whose contiguous node ranges cut few edges (planted communities, ids NOT shuffled): the
    SpMM exchange must take the halo path (no all-gather at all) and match the full-graph oracle."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), SGF_DIST_REORDER="0",   # the caller's order HAS the locality
                      SGF_DIST_OVERLAP=overlap)   # "1": own-column entries multiplied while the halo rows travel
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(2)
        from oracle import sgformer_oracle as O
        from sgformer_amd import ops, synth
        from sgformer_amd.dist import ShardContext, shard_model, sharded_nll_loss
        from sgformer_amd.ours import SGFormer
        from tests.cpu_kernels import CpuKernels
        from tests.test_host import CONFIGS

        ops.set_kernels(CpuKernels())
        cfg = CONFIGS["products"]
        n, f, d, c = 601, 10, 16, 4
        torch.manual_seed(5)
        x = torch.randn(n, f)
        ei = synth.synthetic_graph_community(n, 8.0, seed=3, comm_size=(20, 40), comms_per_super=4, p_comm=0.9,
                                             p_super=0.09, shuffle_ids=False)
        if directed:   # drop one direction of a third of the pairs: A != A^T, the backward needs its own halo
            keep = (ei[0] <= ei[1]) | (torch.arange(ei.shape[1]) % 3 != 0)
            ei = ei[:, keep]
        y = torch.randint(0, c, (n,))
        idx = torch.randperm(n)[: n // 2]
        p = O.init_params(cfg, f, d, c, seed=6)
        ctx = ShardContext(n)
        m = SGFormer(f, d, c, trans_dropout=0.0, gnn_dropout=0.0, **cfg)
        m.load_state_dict({**m.state_dict(), **p})
        shard_model(m, ctx)
        m.train()
        logits = m(ctx.shard_rows(x), ei)
        loss = sharded_nll_loss(logits, ctx.shard_rows(y), ctx.local_index(idx), idx.numel())
        loss.backward()
        ctx.sync_grads(m.parameters())
        p64 = {k: v.double().requires_grad_(v.is_floating_point() and "running" not in k) for k, v in p.items()}
        ref = O.sgformer_forward(p64, x.double(), ei, cfg, training=True)
        O.nll_loss(ref, y, idx).backward()
        gmax = max(float(v.grad.norm()) for v in p64.values() if v.grad is not None)
        gerr = 0.0
        for k, prm in m.named_parameters():
            if p64[k].grad is not None:
                e = float((prm.grad.double() - p64[k].grad).norm())
                gerr = max(gerr, e / (float(p64[k].grad.norm()) + 1e-3 * gmax))
        g = ctx.graph_for(ei)
        ret[rank] = {"logits": float((logits.detach().double() - ref.detach()[ctx.r0:ctx.r1]).abs().max()),
                     "grad": gerr, "halo_sent": ctx.bytes_halo_sent, "gathered": ctx.bytes_all_gathered,
                     "n_halo": g.halo(ctx, False).n_halo, "fraction": g.halo(ctx, False).max_fraction,
                     "symmetric": bool(g.symmetric), "plans": len(g._halo),
                     "split": g.halo(ctx, False)._split is not None, "overlap": ctx.overlap}
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,directed,overlap", [(2, False, "1"), (3, False, "1"), (2, True, "1"), (2, False, "0"),
                                                    (3, True, "0")])
def test_halo_exchange_matches_full_graph(world, directed, overlap):
    """halo all-gather for cut-edge neighbour features (large/ours.py:34 sharded by rows); with the
    own-column entries multiplied while the halo rows are on the links (default) and as one product after the exchange."""
    port = _free_port()
    ret = mp.Manager().dict()
    mp.spawn(_worker_halo, args=(world, port, directed, ret, overlap), nprocs=world, join=True)
    assert len(ret) == world
    for rank in range(world):
        e = ret[rank]
        assert e["overlap"] == (overlap == "1") and e["split"] == (overlap == "1"), e
        assert e["logits"] < 5e-5 and e["grad"] < 2e-3, e
        assert e["gathered"] == 0 and e["halo_sent"] > 0, e          # halo path only: no all-gather of X
        assert e["fraction"] <= 0.5 and 0 < e["n_halo"] < 601 // world, e
        assert e["symmetric"] != directed and e["plans"] == (2 if directed else 1), e


def _worker_local_edges(rank, world, port, ret):
    """Weak-scaling mode: every rank generates ONLY its own rows of the graph (synthetic_graph_shard) and
    hands `model` those local edges; the union over ranks is the reference graph."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), SGF_DIST_CHUNK_COLS="4")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(2)
        from oracle import sgformer_oracle as O
        from sgformer_amd import ops, synth
        from sgformer_amd.dist import ShardContext, ShardedGraph, shard_model, sharded_nll_loss
        from sgformer_amd.ours import SGFormer
        from tests.cpu_kernels import CpuKernels
        from tests.test_host import CONFIGS

        ops.set_kernels(CpuKernels())
        cfg = CONFIGS["products"]
        n_per, f, d, c = 67, 10, 16, 4
        n = n_per * world
        ei_local = synth.synthetic_graph_shard(n_per, 6.0, rank, world, seed=9)
        parts = [None] * world
        dist.all_gather_object(parts, ei_local)
        ei = torch.cat(parts, dim=1)                      # the global graph, for the reference only
        errs = {"targets_local": bool(((ei_local[1] >= rank * n_per) & (ei_local[1] < (rank + 1) * n_per)).all())}
        key = ei[0] * n + ei[1]
        errs["symmetric"] = bool(torch.equal(torch.sort(key)[0], torch.sort(ei[1] * n + ei[0])[0]))
        errs["coalesced"] = bool(torch.unique(key).numel() == key.numel())
        errs["self_loops"] = int((ei[0] == ei[1]).sum()) == n

        torch.manual_seed(5)
        x = torch.randn(n, f)
        y = torch.randint(0, c, (n,))
        idx = torch.randperm(n)[: n // 2]
        p = O.init_params(cfg, f, d, c, seed=6)

        ctx = ShardContext(n, local_edges=True)
        # the block built from local edges alone == the block cut out of the full CSR
        g_loc = ShardedGraph(ei_local, ctx)
        g_ref = ShardedGraph(ei, ShardContext(n))
        errs["csr_bit_exact"] = bool(torch.equal(g_loc.rowptr, g_ref.rowptr) and torch.equal(g_loc.colind, g_ref.colind)
                                     and torch.equal(g_loc.val, g_ref.val))

        m = SGFormer(f, d, c, trans_dropout=0.0, gnn_dropout=0.0, **cfg)
        m.load_state_dict({**m.state_dict(), **p})
        shard_model(m, ctx)
        m.train()
        logits = m(ctx.shard_rows(x), ei_local)
        loss = sharded_nll_loss(logits, ctx.shard_rows(y), ctx.local_index(idx), idx.numel())
        loss.backward()
        ctx.sync_grads(m.parameters())
        total = loss.detach().clone()
        dist.all_reduce(total)

        p64 = {k: v.double().requires_grad_(v.is_floating_point() and "running" not in k) for k, v in p.items()}
        ref = O.sgformer_forward(p64, x.double(), ei, cfg, training=True, bn_stats={})
        lref = O.nll_loss(ref, y, idx)
        lref.backward()
        errs["logits"] = float((logits.detach().double() - ref.detach()[ctx.r0:ctx.r1]).abs().max())
        errs["loss"] = abs(float(total) - float(lref))
        gmax = max(float(v.grad.norm()) for v in p64.values() if v.grad is not None)
        gerr = 0.0
        for k, prm in m.named_parameters():
            if p64[k].grad is not None:
                e = float((prm.grad.double() - p64[k].grad).norm())
                gerr = max(gerr, e / (float(p64[k].grad.norm()) + 1e-3 * gmax))
        errs["grad"] = gerr

        # contract violations fail loudly: a foreign target, an asymmetric graph
        bad = ei_local.clone()
        if rank == world - 1:   # on ONE rank only: the others must raise too (no rank left in a collective)
            bad[1, 0] = (int(bad[1, 0]) + n_per) % n
        try:
            ShardedGraph(bad, ctx)
            errs["foreign_target_raises"] = False
        except ValueError:
            errs["foreign_target_raises"] = True
        asym = ei_local
        if rank == 0:   # drop one off-diagonal edge on one rank only
            off = (ei_local[0] != ei_local[1]).nonzero()[0, 0]
            asym = torch.cat([ei_local[:, :off], ei_local[:, off + 1:]], dim=1)
        try:
            ShardedGraph(asym, ctx)
            errs["asymmetric_raises"] = False
        except ValueError:
            errs["asymmetric_raises"] = True
        # a directed RING over all nodes (+ self-loops): every node has equal in- and out-degree, so any
        # LINEAR checksum of (src, dst) passes it; the per-edge hash must not
        ring_t = torch.arange(ctx.r0, ctx.r1)
        ring = torch.stack([torch.cat([(ring_t + 1) % n, ring_t]), torch.cat([ring_t, ring_t])])
        try:
            ShardedGraph(ring, ctx)
            errs["directed_ring_raises"] = False
        except ValueError:
            errs["directed_ring_raises"] = True
        ret[rank] = errs
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_local_edge_shards_match_full_graph(world):
    """BASELINE.json config 5 (weak scaling): no rank ever sees the global edge list."""
    port = _free_port()
    ret = mp.Manager().dict()
    mp.spawn(_worker_local_edges, args=(world, port, ret), nprocs=world, join=True)
    assert len(ret) == world
    for rank in range(world):
        e = ret[rank]
        for flag in ("targets_local", "symmetric", "coalesced", "self_loops", "csr_bit_exact",
                     "foreign_target_raises", "asymmetric_raises", "directed_ring_raises"):
            assert e[flag], (flag, e)
        assert e["logits"] < 5e-5, e
        assert e["loss"] < 1e-5, e
        assert e["grad"] < 2e-3, e


def _worker_bench_inputs(rank, world, port, workload, ret):
    """bench.py's own input builder + step sequence (make_inputs -> shard_model -> sharded loss ->
    sync_grads) under gloo with the CPU kernel table: what `bench.py --gpus N` runs on RCCL."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(2)
        import bench
        from sgformer_amd import ops
        from sgformer_amd.dist import shard_model, sharded_nll_loss
        from sgformer_amd.ours import SGFormer
        from tests.cpu_kernels import CpuKernels

        ops.set_kernels(CpuKernels())
        n, f, c, d, cfg, weak, ei, x, y, idx, n_train, ctx = bench.make_inputs(workload, 90, 3, rank, world, "cpu")
        torch.manual_seed(3)
        m = SGFormer(f, 16, c, trans_dropout=0.0, gnn_dropout=0.0, **cfg)
        shard_model(m, ctx)
        m.train()
        logits = m(x, ei)
        loss = sharded_nll_loss(logits, y, idx, n_train)
        loss.backward()
        ctx.sync_grads(m.parameters())
        total = loss.detach().clone()
        dist.all_reduce(total)
        g0 = torch.cat([p.grad.reshape(-1) for p in m.parameters() if p.grad is not None])
        gs = [None] * world
        dist.all_gather_object(gs, g0)
        ret[rank] = {"weak": weak, "n": n, "rows": x.shape[0], "logit_rows": logits.shape[0], "loss": float(total),
                     "n_train": n_train, "local_edges": ctx.local_edges,
                     "targets_local": bool(((ei[1] >= ctx.r0) & (ei[1] < ctx.r1)).all()),
                     "grads_equal": all(torch.equal(g, gs[0]) for g in gs)}
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("workload", ["papers100M-weak", "ogbn-products"])
def test_bench_inputs_and_step_under_gloo(workload):
    world = 2
    port = _free_port()
    ret = mp.Manager().dict()
    mp.spawn(_worker_bench_inputs, args=(world, port, workload, ret), nprocs=world, join=True)
    for rank in range(world):
        e = ret[rank]
        weak = workload.endswith("-weak")
        assert e["weak"] == weak and e["local_edges"] == weak
        assert e["n"] == (90 * world if weak else 90)            # weak: N grows with the world
        assert e["rows"] == e["logit_rows"] == (90 if weak else 45)
        assert e["n_train"] == (90 if weak else 45)              # GLOBAL count of training rows
        if weak:
            assert e["targets_local"]
        assert e["grads_equal"] and e["loss"] == e["loss"] and 0 < e["loss"] < 50
    assert ret[0]["loss"] == ret[1]["loss"]


def _worker_bf16(rank, world, port, ret):
    """bf16 storage: the square Linear layers, the two-operand Linear and both stems take the streaming row kernels,
    whose BatchNorm sums are all-reduced INSIDE those ops (ops._linear_with_stats / _StemPair)."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(2)
        from oracle import sgformer_oracle as O
        from sgformer_amd import ops, synth
        from sgformer_amd.dist import ShardContext, shard_model, sharded_nll_loss
        from sgformer_amd.ours import SGFormer
        from tests.cpu_kernels import CpuKernels

        ops.set_kernels(CpuKernels())
        cfg = dict(synth.RECIPES["ogbn-products"])
        n, f, d, c = 403, 12, 64, 4
        torch.manual_seed(5)
        x = torch.randn(n, f)
        ei = O.synthetic_graph(n, 6.0, seed=4)
        y = torch.randint(0, c, (n,))
        idx = torch.randperm(n)[: n // 2]

        def build():
            torch.manual_seed(17)
            return SGFormer(f, d, c, trans_dropout=0.0, gnn_dropout=0.0, compute_dtype=torch.bfloat16, **cfg).train()

        calls = {"stem": 0, "cat": 0}
        for name, key in (("stem_pair", "stem"), ("gcn_epilogue_cat", "cat")):
            orig = getattr(CpuKernels, name)
            setattr(CpuKernels, name, staticmethod(
                lambda *a, _o=orig, _k=key, **k: (calls.__setitem__(_k, calls[_k] + 1), _o(*a, **k))[1]))
        ctx = ShardContext(n)
        m = build()
        shard_model(m, ctx)
        logits = m(ctx.shard_rows(x), ei)
        loss = sharded_nll_loss(logits, ctx.shard_rows(y), ctx.local_index(idx), idx.numel())
        loss.backward()
        ctx.sync_grads(m.parameters())
        sharded_calls = dict(calls)

        ref_m = build()                                   # the same model, one process, same kernel table
        ref = ref_m(x, ei)
        torch.nn.functional.nll_loss(torch.log_softmax(ref.float(), 1)[idx], y[idx]).backward()
        errs = {"logits": float((logits.detach().float() - ref.detach().float()[ctx.r0:ctx.r1]).abs().max()),
                "scale": float(ref.detach().float().abs().max()), "calls": sharded_calls}
        gerr = 0.0
        rg = dict(ref_m.named_parameters())
        for k, prm in m.named_parameters():
            if prm.grad is None or (k.startswith("graph_conv") and k.endswith("bias") and ("fcs.0" in k or ".W." in k)):
                continue
            den = float(rg[k].grad.double().norm())
            gerr = max(gerr, float((prm.grad.double() - rg[k].grad.double()).norm()) / max(den, 1e-6))
        errs["grad"] = gerr
        sd, rsd = m.state_dict(), ref_m.state_dict()
        errs["running_var"] = max(float((sd[k] - rsd[k]).abs().max() / rsd[k].abs().max()) for k in sd if "running_var" in k)
        ret[rank] = errs
    finally:
        dist.destroy_process_group()


def test_sharded_bf16_streaming_paths_match_single_process():
    world = 2
    ret = mp.Manager().dict()
    mp.spawn(_worker_bf16, args=(world, _free_port(), ret), nprocs=world, join=True)
    for rank in range(world):
        e = ret[rank]
        assert e["calls"]["stem"] >= 2 and e["calls"]["cat"] >= 3, e
        assert e["logits"] <= 3e-2 * max(1.0, e["scale"]), e
        assert e["grad"] <= 0.2, e
        assert e["running_var"] <= 2e-2, e


def _worker_repartition(rank, world, port, reorder, directed, ret):
    """A community graph whose ids are SHUFFLED: contiguous ranges of the caller's numbering cut almost every edge.
    With SGF_DIST_REORDER=1 (default) every rank computes the same sgf_reorder permutation, the ranks re-partition in
    that order (features in / logits out through one all-to-all each) and the SpMM takes the halo path; with 0 it falls
    back to the all-gather.  Either way the result is the full-graph oracle's in the CALLER's order."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), SGF_DIST_REORDER=reorder, SGF_DIST_CHUNK_COLS="64")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(2)
        from oracle import sgformer_oracle as O
        from sgformer_amd import ops, synth
        from sgformer_amd.dist import ShardContext, shard_model, sharded_nll_loss
        from sgformer_amd.ours import SGFormer
        from tests.cpu_kernels import CpuKernels
        from tests.test_host import CONFIGS

        ops.set_kernels(CpuKernels())
        cfg = CONFIGS["products"]
        n, f, d, c = 611, 10, 16, 4
        torch.manual_seed(5)
        x = torch.randn(n, f)
        ei = synth.synthetic_graph_community(n, 8.0, seed=3, comm_size=(20, 40), comms_per_super=4, p_comm=0.92,
                                             p_super=0.07, shuffle_ids=True)
        if directed:
            keep = (ei[0] <= ei[1]) | (torch.arange(ei.shape[1]) % 3 != 0)
            ei = ei[:, keep]
        y = torch.randint(0, c, (n,))
        idx = torch.randperm(n)[: n // 2]
        p = O.init_params(cfg, f, d, c, seed=6)
        ctx = ShardContext(n)
        m = SGFormer(f, d, c, trans_dropout=0.0, gnn_dropout=0.0, **cfg)
        m.load_state_dict({**m.state_dict(), **p})
        shard_model(m, ctx)
        m.train()
        xl = ctx.shard_rows(x).contiguous()
        for _ in range(2):                                 # second step: the re-partitioned features are reused
            m.zero_grad(set_to_none=True)
            logits = m(xl, ei)
            loss = sharded_nll_loss(logits, ctx.shard_rows(y), ctx.local_index(idx), idx.numel())
            loss.backward()
        ctx.sync_grads(m.parameters())
        p64 = {k: v.double().requires_grad_(v.is_floating_point() and "running" not in k) for k, v in p.items()}
        ref = O.sgformer_forward(p64, x.double(), ei, cfg, training=True)
        O.nll_loss(ref, y, idx).backward()
        gmax = max(float(v.grad.norm()) for v in p64.values() if v.grad is not None)
        gerr = 0.0
        for k, prm in m.named_parameters():
            if p64[k].grad is not None:
                e = float((prm.grad.double() - p64[k].grad).norm())
                gerr = max(gerr, e / (float(p64[k].grad.norm()) + 1e-3 * gmax))
        rp = ctx.repartition_for(ei)
        ret[rank] = {"logits": float((logits.detach().double() - ref.detach()[ctx.r0:ctx.r1]).abs().max()), "grad": gerr,
                     "halo_sent": ctx.bytes_halo_sent, "gathered": ctx.bytes_all_gathered,
                     "repartition": ctx.bytes_repartition, "adopted": rp is not None,
                     "fraction": rp.stats["halo_fraction"] if rp is not None else None}
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,reorder,directed", [(2, "1", False), (3, "1", False), (2, "1", True), (2, "0", False)])
def test_repartition_in_reorder_order_engages_the_halo_path(world, reorder, directed):
    """a locality order across ranks, so that the halo exchange serves graphs whose given ids
    carry no locality (large/ours.py:34 sharded by rows)."""
    port = _free_port()
    ret = mp.Manager().dict()
    mp.spawn(_worker_repartition, args=(world, port, reorder, directed, ret), nprocs=world, join=True)
    assert len(ret) == world
    for rank in range(world):
        e = ret[rank]
        assert e["logits"] < 5e-5 and e["grad"] < 2e-3, e
        if reorder == "1":
            assert e["adopted"] and e["fraction"] <= 0.5, e
            assert e["gathered"] == 0 and e["halo_sent"] > 0 and e["repartition"] > 0, e
        else:
            assert not e["adopted"] and e["gathered"] > 0 and e["repartition"] == 0, e


def _worker_batch(rank, world, port, ret):
    """BASELINE.json config 5's mode (SURVEY.md §8e last row): every rank trains on a mini-batch drawn from ITS node
    shard — own induced subgraph, no halo — while the attention set is the union of the ranks' batches (N = the global
    batch size in num / den, K^T V and BatchNorm partial sums all-reduced).  Reference: ONE process on the concatenated
    batch with the block-diagonal union of the subgraphs."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(2)
        from oracle import sgformer_oracle as O
        from sgformer_amd import ops
        from sgformer_amd.dist import ShardContext, shard_model, sharded_nll_loss
        from sgformer_amd.ours_100m import SGFormer
        from tests.cpu_kernels import CpuKernels

        ops.set_kernels(CpuKernels())
        cfg = dict(alpha=0.5, trans_num_layers=1, gnn_num_layers=2, gnn_use_init=True, graph_weight=0.8)
        f, d, c = 10, 16, 5
        sizes = [57, 44, 63][:world]                        # ragged batches
        xs, eis, ys, idxs = [], [], [], []
        for r, nb in enumerate(sizes):                       # every rank builds ALL batches: the reference needs them
            g = torch.Generator().manual_seed(100 + r)
            xs.append(torch.randn(nb, f, generator=g))
            eis.append(O.synthetic_graph(nb, 4.0, seed=20 + r, directed=True))       # sampled batches are directed
            ys.append(torch.randint(0, c, (nb,), generator=g))
            idxs.append(torch.randperm(nb, generator=g)[: nb // 2])
        p = O.init_params(cfg, f, d, c, seed=6)
        ctx = ShardContext.for_batch(sizes[rank])
        assert ctx.n_global == sum(sizes) and ctx.r0 == sum(sizes[:rank]) and ctx.local_graph
        m = SGFormer(f, d, c, trans_dropout=0.0, gnn_dropout=0.0, **cfg)
        m.load_state_dict({**m.state_dict(), **p})
        shard_model(m, ctx)
        m.train()
        n_train = sum(int(i.numel()) for i in idxs)
        logits = m(xs[rank], eis[rank])
        loss = sharded_nll_loss(logits, ys[rank], idxs[rank], n_train)
        loss.backward()
        ctx.sync_grads(m.parameters())
        total = loss.detach().clone()
        dist.all_reduce(total)
        # reference: one process, concatenated batch, block-diagonal graph
        off = [sum(sizes[:r]) for r in range(world)]
        xa, ya = torch.cat(xs), torch.cat(ys)
        eia = torch.cat([e + o for e, o in zip(eis, off)], dim=1)
        ida = torch.cat([i + o for i, o in zip(idxs, off)])
        p64 = {k: v.double().requires_grad_(v.is_floating_point() and "running" not in k) for k, v in p.items()}
        ref = O.sgformer_forward(p64, xa.double(), eia, cfg, training=True)
        lref = O.nll_loss(ref, ya, ida)
        lref.backward()
        gmax = max(float(v.grad.norm()) for v in p64.values() if v.grad is not None)
        gerr = 0.0
        for k, prm in m.named_parameters():
            if p64[k].grad is not None:
                e = float((prm.grad.double() - p64[k].grad).norm())
                gerr = max(gerr, e / (float(p64[k].grad.norm()) + 1e-3 * gmax))
        ret[rank] = {"logits": float((logits.detach().double() - ref.detach()[ctx.r0:ctx.r1]).abs().max()),
                     "loss": abs(float(total) - float(lref)), "grad": gerr, "gathered": ctx.bytes_all_gathered,
                     "halo": ctx.bytes_halo_sent, "reduced": ctx.bytes_all_reduced}
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_minibatch_per_rank_with_global_attention_set(world):
    """SURVEY.md §8e 'mini-batch + 8 GPU (cfg 5)': 100M/nb-sample.py:27-45 per rank, attention over the union batch."""
    port = _free_port()
    ret = mp.Manager().dict()
    mp.spawn(_worker_batch, args=(world, port, ret), nprocs=world, join=True)
    assert len(ret) == world
    for rank in range(world):
        e = ret[rank]
        assert e["logits"] < 5e-5 and e["loss"] < 1e-5 and e["grad"] < 2e-3, e
        assert e["gathered"] == 0 and e["halo"] == 0 and e["reduced"] > 0, e      # no operand crosses ranks


@pytest.mark.parametrize("graph", ["uniform", "community"])
def test_bench_under_the_drivers_launch_line(graph, tmp_path):
    """`python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P bench.py
    --gpus 2 --steps K --warmup W` — EXACTLY what the driver runs for the scaling bench — on this GPU-less host:
    SGF_BENCH_DRYRUN=1 swaps RCCL for gloo and libsgf.so for the CPU kernel table and leaves the rest of bench.py as it
    is: env parsing, rendezvous, input sharding, step sequence, the fenced timing + max over ranks, one JSON line from
    rank 0 with the contract's fields.  `community`: the graph whose locality sgf_reorder recovers — the partition then
    follows it across ranks (dist.Repartition) and the SpMM runs the halo exchange instead of the all-gather."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {**os.environ, "SGF_BENCH_DRYRUN": "1", "OMP_NUM_THREADS": "2"}
    if graph == "community":
        env["SGF_HALO_MAX"] = "1.0"          # 4000 nodes are ONE super-community: take the halo path whatever its size
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--nodes", "4000", "--graph", graph]
    p = subprocess.run(cmd, cwd=tmp_path, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]                       # rank 0 only
    out = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config"):
        assert key in out, key
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["warmup"] == 1 and out["config"]["dry_run"] is True
    assert out["scaling"] == "strong" and out["config"]["parallelism"] == "node-shard x2"
    assert abs(out["value"] - 4000 / (out["ms_per_step"] * 1e-3)) <= 1e-6 * out["value"]
    ex = out["config"]["exchanged"]
    assert ex["all_reduce_bytes_per_step"] > 0
    if graph == "community":
        assert ex["halo_bytes_sent_per_step"] > 0 and ex["all_gather_bytes_per_step"] == 0 and ex["repartition_bytes_per_step"] > 0
    else:
        assert ex["all_gather_bytes_per_step"] > 0
    assert out["loss"] == out["loss"] and 0.0 < out["loss"] < 20.0


@pytest.mark.parametrize("workload,world", [("pokec", 2), ("papers100M-weak", 2)])
def test_bench_dry_run_of_the_multi_gpu_configs_prints_exchange_bytes_and_a_labelled_model(workload, world, tmp_path):
    """BASELINE configs 4 (pokec, node-sharded K^T V all-reduce) and 5 (papers100M-shaped, weak scaling) through the driver's
    launch line on this GPU-less host (gloo + CPU kernel table, toy node count): the line carries the bytes every exchange
    moved per step AND a `scaling_model` object that says of itself that it is a model — no multi-GPU number in this
    repository is a measurement."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {**os.environ, "SGF_BENCH_DRYRUN": "1", "OMP_NUM_THREADS": "2"}
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", str(world), "--steps", "1",
           "--warmup", "1", "--nodes", "600", "--workload", workload]
    p = subprocess.run(cmd, cwd=tmp_path, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    out = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][0])
    assert out["n_gpus"] == world and out["config"]["dry_run"] is True
    assert out["scaling"] == ("weak" if workload.endswith("-weak") else "strong")
    ex = out["config"]["exchanged"]
    assert ex["all_reduce_bytes_per_step"] > 0 and (ex["all_gather_bytes_per_step"] > 0 or ex["halo_bytes_sent_per_step"] > 0
                                                     or workload.endswith("-weak"))
    m = out["scaling_model"]
    assert m["label"].startswith("MODEL") and "not a measurement" in m["label"]
    assert m["all_gather_bytes_received_per_rank_per_step"] > 0
    # the model starts from THIS run's step time, not from a constant of an earlier round
    assert abs(m["measured_step_ms"] - out["ms_per_step"]) < 1e-2 and "implied_compute_ms_per_rank" in m
    assert "modelled_nodes_per_s" not in m and "modelled_efficiency" not in m


def test_the_drivers_8_rank_line_for_config_5_and_its_per_rank_memory(tmp_path):
    """the driver's own launch line for BASELINE config 5 — `python -m torch.distributed.run --nnodes=1
    --nproc-per-node 8 ... bench.py --gpus 8 --workload papers100M-weak` — on this GPU-less host (dry run: gloo, CPU kernel
    table, toy node count): eight ranks rendezvous, every rank generates only its own rows (synth.synthetic_graph_shard), the
    line says which way the SpMM operand travelled, carries no scaling number that is not labelled a model, and the per-rank
    device memory AT FULL SIZE (111 M nodes over 8 GPUs), by formula, stays under the 288 GB of one MI355X."""
    import json
    import subprocess
    import sys
    from benchlib.model import per_rank_memory_model
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {**os.environ, "SGF_BENCH_DRYRUN": "1", "OMP_NUM_THREADS": "1"}
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "1",
           "--nodes", "300", "--workload", "papers100M-weak"]
    p = subprocess.run(cmd, cwd=tmp_path, env=env, capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    out = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][0])
    assert out["n_gpus"] == 8 and out["scaling"] == "weak" and out["config"]["dry_run"] is True
    assert out["config"]["nodes"] == 8 * 300 and "nnz_per_rank" in out["config"]
    ex = out["config"]["exchanged"]
    assert ex["spmm_exchange"]["path"] in ("halo", "all_gather") and 0.0 <= ex["spmm_exchange"]["halo_fraction_max"] <= 1.0
    assert ex["all_reduce_bytes_per_step"] > 0 and "halo_exchanges_overlapped_per_step" in ex
    assert out["scaling_model"]["label"].startswith("MODEL") and out["per_rank_memory_model"]["label"].startswith("MODEL")
    assert "efficiency" not in json.dumps(out)                    # the driver computes efficiency itself
    # the full-size shape by formula: every component, their sum, and the 288 GB bound
    m = per_rank_memory_model("papers100M-weak", 8, "bf16")
    assert m["per_rank_rows"] == 13882494 and m["per_rank_rows"] * 8 >= 111_059_952
    assert set(m["bytes"]) == {"features", "edges", "csr", "activations", "exchange"}
    assert m["total_bytes"] == sum(m["bytes"].values()) and m["total_bytes"] < 288e9 and m["fits"]
    assert m["bytes"]["exchange"] == (8 - 1) * 13882494 * 128 * 2
    # the anchor: the measured peak of ONE GPU's share (profiles/r05_bench_p100.json: 98.49 GB) lies under the formula's figure
    one = per_rank_memory_model("papers100M-shard8", 1, "bf16")
    assert 98.49e9 < one["total_bytes"] < 1.35 * 98.49e9
