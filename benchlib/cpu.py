"""bench.py: `cpu_baseline` — the oracle (test infrastructure) timed on the host cores of the box, never a fallback."""
from __future__ import annotations

import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from sgformer_amd import ops, synth  # noqa: E402

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
