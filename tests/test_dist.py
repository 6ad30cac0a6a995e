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
