"""bench.py's two NON-measurement modes (moved out of bench.py in r05: test code does not belong in the bench).

  SGF_BENCH_DRYRUN=1     tests/test_dist.py only — the driver's launch line on a GPU-less host;
  SGF_BENCH_SHARE_GPU=1  validation on a 1-GPU box — N ranks on cuda:0 with a host-staged gloo transport.
Lines printed in either mode are marked (`dry_run` / `shared_gpu`) and are never measurements."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def enter_dryrun(bench):
    """SGF_BENCH_DRYRUN=1 (tests/test_dist.py only; never a measurement): the driver's launch line
    `python -m torch.distributed.run ... bench.py --gpus N ...` on a GPU-less host — gloo instead of RCCL, the CPU kernel
    table of tests/cpu_kernels.py instead of libsgf.so, torch.cuda's fences as no-ops.  What it exercises is everything
    of this file that is not a kernel: env parsing, rendezvous, sharding of the inputs, the step sequence, the timing
    fences and the max-over-ranks reduction, the JSON contract.  The printed line is marked `dry_run`."""
    sys.path.insert(0, os.path.join(ROOT))
    from tests.cpu_kernels import CpuKernels
    bench.ops.set_kernels(CpuKernels())
    for name in ("synchronize", "set_device", "empty_cache", "reset_peak_memory_stats"):
        setattr(torch.cuda, name, lambda *a, **k: None)
    torch.cuda.max_memory_allocated = lambda *a, **k: 0
    bench.SpmmTimer.install = lambda self: None
    bench.SpmmTimer.uninstall = lambda self: None


def enter_shared_gpu():
    """SGF_BENCH_SHARE_GPU=1 (validation on a 1-GPU box; never a measurement): the N ranks of the driver's launch line all
    run on cuda:0 — the node-sharded step with the REAL kernels of libsgf.so (own-column tile / stream SpMM, halo placement,
    sgf_gather_rows packing, SyncBN statistics, fused Adam) and the real step sequence — while the transport is gloo with the
    collective's tensors staged through the host (RCCL refuses two ranks on one device).  The printed line is marked
    `shared_gpu`; its loss must equal the one-rank run's."""
    real = {k: getattr(dist, k) for k in ("all_reduce", "broadcast", "all_to_all_single", "all_gather_into_tensor")}

    class _Done:
        def wait(self, *a, **k):
            return True

        def is_completed(self):
            return True

    def staged(name, outs, ins):
        def call(*args, async_op=False, **kw):
            args = list(args)
            dev_t = {i: args[i] for i in set(outs) | set(ins) if i < len(args) and torch.is_tensor(args[i]) and args[i].is_cuda}
            for i, t in dev_t.items():
                args[i] = t.cpu() if i in ins else torch.empty(t.shape, dtype=t.dtype)
            real[name](*args, **kw)
            for i, t in dev_t.items():
                if i in outs:
                    t.copy_(args[i])
            return _Done() if async_op else None
        return call
    dist.all_reduce = staged("all_reduce", outs=(0,), ins=(0,))
    dist.broadcast = staged("broadcast", outs=(0,), ins=(0,))
    dist.all_to_all_single = staged("all_to_all_single", outs=(0,), ins=(1,))
    dist.all_gather_into_tensor = staged("all_gather_into_tensor", outs=(0,), ins=(1,))


