#!/usr/bin/env python
"""Summarise the rocprofv3 passes written by scripts/pmc_passes.sh: per kernel name, the mean of every
counter over its dispatches and the mean duration from the kernel trace.  Prints a markdown table."""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name):
    for key in ("k_spmm_blk", "k_spmm_wave", "k_spmm_sub", "k_spmm_long_seg", "k_spmm_long_fin"):
        if key in name:
            return key + ("<bf16>" if "It" in name.split(key)[1][:4] else "<f32>")
    return None


def main(out):
    counters = defaultdict(lambda: defaultdict(list))      # kernel -> counter -> values
    for f in glob.glob(os.path.join(out, "p*", "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = short(row.get("Kernel_Name", ""))
            if k:
                # LDS size distinguishes block shapes of the row-block kernel
                k = f"{k} lds={row.get('LDS_Block_Size', '?')}" if "blk" in k else k
                counters[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    dur = defaultdict(list)
    for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_trace.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = short(row.get("Kernel_Name", ""))
            if k:
                k = f"{k} lds={row.get('LDS_Block_Size', '?')}" if "blk" in k else k
                dur[k].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6)
    names = sorted({c for k in counters for c in counters[k]})
    print("| kernel | launches | mean ms | " + " | ".join(names) + " |")
    print("|---|---:|---:|" + "---:|" * len(names))
    for k in sorted(set(counters) | set(dur)):
        d = dur.get(k, [])
        cells = []
        for c in names:
            v = counters[k].get(c, [])
            cells.append(f"{sum(v) / len(v):.4g}" if v else "")
        print(f"| {k} | {len(d)} | {sum(d) / len(d):.3f} | " + " | ".join(cells) + " |" if d else
              f"| {k} | 0 | | " + " | ".join(cells) + " |")


if __name__ == "__main__":
    main(sys.argv[1])
