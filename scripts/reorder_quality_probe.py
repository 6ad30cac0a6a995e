#!/usr/bin/env python
"""How much of the generator's planted two-level structure does sgf_reorder's numbering expose?  (probe; GPU)

Prints, for the re-ordered community graph: the share of stored entries whose |row - column| is below a set of windows, next
to the planted ideal (same community 80 %, same super-community 95 %), and the sizes of the level-2 groups."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sgformer_amd import ops, synth  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    n, deg = 2449029, 50.5
    which = sys.argv[1] if len(sys.argv) > 1 else "community"
    if which == "community":
        ei, labels = synth.synthetic_graph_community(n, deg, seed=123, device=dev, return_labels=True)
    else:
        ei = synth.synthetic_graph_community_powerlaw(n, deg, seed=123, device=dev)
        labels = None
    perm, inv, comm = ops.K.reorder(ei, n, *ops.REORDER_ITERS)
    pos = inv.long()
    d = (pos[ei[0]] - pos[ei[1]]).abs()
    res = {"graph": which, "iters": list(ops.REORDER_ITERS)}
    for w in (128, 256, 1024, 4096, 8192, 16384, 32768, 65536, 262144):
        res[f"within_{w}"] = round(float((d < w).float().mean()), 4)
    cs = comm[perm.long()]                                     # level-1 community of every position
    res["communities"] = int(torch.unique(cs).numel())
    if labels is not None:
        sup = labels // 64
        res["planted_same_comm"] = round(float((labels[ei[0]] == labels[ei[1]]).float().mean()), 4)
        res["planted_same_super"] = round(float((sup[ei[0]] == sup[ei[1]]).float().mean()), 4)
        # positions of one planted super-community: how spread are they in the new order?
        spans = []
        for s in (0, 7, 100, 200):
            p = pos[sup == s]
            q = torch.quantile(p.float(), torch.tensor([0.05, 0.5, 0.95], device=dev))
            spans.append({"super": s, "nodes": int(p.numel()), "p05": int(q[0]), "p50": int(q[1]), "p95": int(q[2]),
                          "distinct_8k_windows": int(torch.unique(p // 8192).numel())})
        res["spans"] = spans
    print(json.dumps(res))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "reorder_quality.jsonl"), "a") as f:
        f.write(json.dumps(res) + "\n")


if __name__ == "__main__":
    main()
