"""Node-sharded step over RCCL (backend "nccl" on ROCm) on REAL GPUs — needs at least two visible devices,
so it is skipped on the 1-GPU test boxes; the same exchange logic is covered on CPU by tests/test_dist.py
(gloo, world size 2-3).  Checks the halo path (planted communities) and the all-gather fallback (uniform
graph) against the fp64 oracle of the full graph (large/ours.py:265-276 end to end)."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, kind, ret):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    try:
        from oracle import sgformer_oracle as O
        from sgformer_amd import synth
        from sgformer_amd.dist import ShardContext, shard_model, sharded_nll_loss
        from sgformer_amd.ours import SGFormer
        cfg = dict(synth.RECIPES["ogbn-products"])
        n, f, d, c = 6001, 24, 64, 5
        torch.manual_seed(5)
        x = torch.randn(n, f)
        if kind == "halo":
            ei = synth.synthetic_graph_community(n, 10.0, seed=3, comm_size=(30, 60), comms_per_super=4, p_comm=0.9,
                                                 p_super=0.09, shuffle_ids=False)
        else:
            ei = synth.synthetic_graph(n, 10.0, seed=3)
        y = torch.randint(0, c, (n,))
        idx = torch.randperm(n)[: n // 2]
        p = O.init_params(cfg, f, d, c, seed=6)
        ctx = ShardContext(n)
        m = SGFormer(f, d, c, trans_dropout=0.0, gnn_dropout=0.0, **cfg)
        m.load_state_dict({**m.state_dict(), **p})
        m = m.to(dev).train()
        shard_model(m, ctx)
        logits = m(ctx.shard_rows(x).to(dev), ei.to(dev))
        loss = sharded_nll_loss(logits, ctx.shard_rows(y).to(dev), ctx.local_index(idx).to(dev), idx.numel())
        loss.backward()
        ctx.sync_grads(m.parameters())
        torch.cuda.synchronize()
        p64 = {k: v.double().requires_grad_(v.is_floating_point() and "running" not in k) for k, v in p.items()}
        ref = O.sgformer_forward(p64, x.double(), ei, cfg, training=True)
        O.nll_loss(ref, y, idx).backward()
        gmax = max(float(v.grad.norm()) for v in p64.values() if v.grad is not None)
        gerr = 0.0
        for k, prm in m.named_parameters():
            if p64[k].grad is not None:
                e = float((prm.grad.double().cpu() - p64[k].grad).norm())
                gerr = max(gerr, e / (float(p64[k].grad.norm()) + 1e-3 * gmax))
        ret[rank] = {"logits": float((logits.detach().double().cpu() - ref.detach()[ctx.r0:ctx.r1]).abs().max()),
                     "grad": gerr, "halo_sent": ctx.bytes_halo_sent, "gathered": ctx.bytes_all_gathered}
    finally:
        dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs >= 2 GPUs (RCCL over xGMI)")
@pytest.mark.parametrize("kind", ["halo", "allgather"])
def test_sharded_step_over_rccl(kind):
    world = 2
    ret = mp.Manager().dict()
    mp.spawn(_worker, args=(world, _free_port(), kind, ret), nprocs=world, join=True)
    for rank in range(world):
        e = ret[rank]
        assert e["logits"] < 1e-4 and e["grad"] < 2e-3, e
        assert (e["halo_sent"] > 0 and e["gathered"] == 0) if kind == "halo" else e["gathered"] > 0, e


@pytest.mark.parametrize("kind", ["halo", "allgather"])
def test_sharded_step_single_rank_over_rccl(kind):
    """The same sharded step with ONE rank: every collective of the path (all-reduce of the attention partials and
    statistics, the halo all_to_all_single / all-gather, the gradient sync) goes through a real RCCL communicator on
    the 1-GPU test box — degenerate exchanges, but the calls, dtypes, devices and split lists are the ones the
    multi-GPU run issues.  Parity as above."""
    ret = mp.Manager().dict()
    mp.spawn(_worker, args=(1, _free_port(), kind, ret), nprocs=1, join=True)
    e = ret[0]
    assert e["logits"] < 1e-4 and e["grad"] < 2e-3, e


def test_bench_under_torchrun_single_rank(tmp_path):
    """bench.py the way the driver launches it for N > 1 (torch.distributed.run, RCCL init, node-sharded model),
    with N = 1 on a small graph: one JSON line with the contract's keys."""
    import json
    import subprocess
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", SGF_BENCH_FORCE_SHARD="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2",
           "--warmup", "1", "--nodes", "20000", "--no-cpu-baseline", "--no-structured"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]
    rec = json.loads(line)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in rec, key
    assert rec["n_gpus"] == 1 and rec["steps"] == 2 and rec["value"] > 0
