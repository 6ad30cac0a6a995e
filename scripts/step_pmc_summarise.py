#!/usr/bin/env python
"""HBM bytes per launch of every kernel of a training step, from two rocprofv3 --pmc passes of bench.py (FETCH_SIZE and
WRITE_SIZE in separate runs, never combined with tracing):

    python scripts/step_pmc_summarise.py gpurun_out/r6_final profiles/r06_products_bf16_kernel_stats.csv > profiles/r06_step_pmc.md

HBM bytes = 2 x FETCH_SIZE KiB + WRITE_SIZE KiB (MI355X_MICROARCH.md §HBM: gfx950 tallies wide coalesced reads at half their
bytes); launch times from the kernel-trace statistics of the same command; only launches over 50 us are listed."""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name):
    for pre in ("void sgf::(anonymous namespace)::", "sgf::(anonymous namespace)::", "void at::native::", "void "):
        name = name.replace(pre, "")
    return name.split("(")[0][:64]


def counter(src, which):
    agg = defaultdict(list)
    for f in glob.glob(os.path.join(src, "pmc_" + which, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == which:
                agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return agg


def main(src, stats_csv):
    fetch, write = counter(src, "FETCH_SIZE"), counter(src, "WRITE_SIZE")
    dur = {short(r["Name"]): (float(r["AverageNs"]) / 1e3, int(r["Calls"])) for r in csv.DictReader(open(stats_csv))}
    print("# HBM traffic per launch of the kernels of one training step (MI355X, bf16, ogbn-products shape, one stream)\n")
    print("`rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` in separate passes over `SGF_OVERLAP=0 python bench.py --graph "
          "uniform --steps 2 --warmup 1 --no-cpu-baseline --no-structured` (scripts/r6_final.sh); HBM bytes = 2 x FETCH_SIZE KiB + "
          "WRITE_SIZE KiB; `avg us` from the kernel-trace statistics of the same command with 5 timed steps.  T = N d s = 1.254 GB.\n")
    print("| kernel | launches in the pass | HBM read GB | HBM write GB | total GB | in T | avg us | TB/s on measured bytes |")
    print("|---|---:|---:|---:|---:|---:|---:|---:|")
    rows = []
    for k in fetch:
        if k not in dur or dur[k][0] < 50:
            continue
        rd = 2 * sum(fetch[k]) / len(fetch[k]) * 1024 / 1e9
        wr = sum(write.get(k, [0])) / max(len(write.get(k, [0])), 1) * 1024 / 1e9
        rows.append((dur[k][0] * dur[k][1], k, len(fetch[k]), rd, wr))
    for _, k, n, rd, wr in sorted(rows, reverse=True):
        us = dur[k][0]
        print(f"| `{k}` | {n} | {rd:.2f} | {wr:.2f} | {rd + wr:.2f} | {(rd + wr) / 1.254:.2f} | {us:.0f} | {(rd + wr) / us * 1e3:.2f} |")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
