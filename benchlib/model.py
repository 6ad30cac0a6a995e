"""bench.py --gpus N: the exchange MODEL printed next to a multi-GPU line (bytes over xGMI links; never a measurement)."""
from __future__ import annotations

from sgformer_amd import synth

XGMI_LINK_GBS = 153.0      # per direction and link, 7 links per GPU (MI355X_MICROARCH.md); 0.8 of it assumed reachable


def scaling_model(workload: str, dtype: str, world: int, measured_step_ms: float):
    """A MODEL of the exchanges of the node-sharded step on `world` GPUs of one node — bytes over xGMI links, NOT a
    measurement — next to the step time THIS run measured (the line's own `ms_per_step`): what is left after the modelled
    exchanges is the compute share the model implies.  No constant of an earlier run enters it.
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
    t_x = (t_all_gather + t_small) * 1e3
    return {"label": "MODEL of the exchanges — not a measurement; the only measured number here is measured_step_ms, this run's own",
            "per_rank_rows": shard_rows, "spmm_launches_per_step": spmm_launches,
            "all_gather_bytes_received_per_rank_per_step": int(spmm_launches * shard_bytes * (world - 1)),
            "all_reduce_bytes_per_step": int(small + n_params * 4),
            "xgmi_link_GBps_assumed": round(XGMI_LINK_GBS * 0.8, 1),
            "t_all_gather_ms_no_overlap": round(t_all_gather * 1e3, 3), "t_small_collectives_ms": round(t_small * 1e3, 3),
            "measured_step_ms": round(measured_step_ms, 3),
            "implied_compute_ms_per_rank": round(measured_step_ms - t_x, 3),
            "note": "exchanges counted WITHOUT overlap (dist.py overlaps the own-column product with the halo exchange, so the "
                    "implied compute share is a lower bound); on a uniform graph the halo is every remote row (all-gather "
                    "fallback), on a graph sgf_reorder can partition the halo plan sends the cut-edge rows only"}


HBM_BYTES = 288e9          # MI355X: 288 GB of HBM3E per GPU


def per_rank_memory_model(workload: str, world: int, dtype: str = "bf16"):
    """Device memory ONE rank of the node-sharded step holds at the workload's full size, by formula (a MODEL: components
    below; the one measured anchor is the 1-GPU share of config 5, 98.5 GB peak in profiles/r05_bench_p100.json against
    114.5 GB by this formula — the allocator re-uses transients the formula counts side by side, so it errs high).
      features        fp32 [n, f] as the trainer hands them in + the module's storage-dtype entry copy
      edges           the rank's edge list, int64 [2, nnz]
      csr             rowptr int64 + colind int32 + val fp32, twice when the rank's block is not symmetric (local_edges shards)
      activations     21 [n, d] tensors alive at the peak of a step (stems 4, attention 2, 3 per GCN layer at Lg = 3, 6 backward
                      transients) + logits and their gradient in fp32
      exchange        node-sharded only: the SpMM operand of every other rank, worst case all of it (all-gather fallback on a
                      graph without locality; the halo plan needs less)"""
    n, deg, f, c, d = synth.SHAPES[workload]
    weak = workload.endswith("-weak")
    cfg = synth.RECIPES.get(workload, synth.RECIPES["ogbn-products"])
    s = 4 if dtype == "f32" else 2
    n_total = n * world if weak else n
    rows = n_total // world
    nnz = int(rows * (deg + 1))
    lg = cfg["gnn_num_layers"]
    T = rows * d * s
    comp = {"features": rows * f * 4 + rows * ((f + 7) // 8 * 8) * s,
            "edges": 2 * nnz * 8,
            "csr": ((rows + 1) * 8 + nnz * 8) * (2 if world > 1 else 1),
            "activations": (6 + 3 * lg + 6) * T + 2 * rows * c * 4,
            "exchange": (n_total - rows) * d * s if world > 1 else 0}
    total = sum(comp.values())
    return {"label": "MODEL by formula — not a measurement", "per_rank_rows": rows, "per_rank_nnz": nnz,
            "bytes": comp, "total_bytes": int(total), "hbm_bytes": int(HBM_BYTES), "fits": total < HBM_BYTES}
