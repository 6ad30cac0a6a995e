#!/usr/bin/env python
"""Per-kernel totals of one rocprofv3 --kernel-trace --stats run: python scripts/trace_top.py <dir> [steps] -> name, calls,
total ms, ms per step, average us (sorted by total).  Reads *_kernel_stats.csv wherever it lies under <dir>."""
import csv
import glob
import sys


def short(name):
    for pre in ("void sgf::(anonymous namespace)::", "sgf::(anonymous namespace)::", "void at::native::", "void "):
        name = name.replace(pre, "")
    return name[:110]


def main():
    d = sys.argv[1]
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 13
    files = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)
    rows = []
    for f in files:
        for r in csv.DictReader(open(f)):
            rows.append((short(r["Name"]), int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
    rows.sort(key=lambda r: -r[2])
    tot = sum(r[2] for r in rows)
    print(f"total {tot:.1f} ms over {steps} steps = {tot / steps:.2f} ms/step; launches/step {sum(r[1] for r in rows) / steps:.1f}")
    for n, c, t, a in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 45]:
        print(f"{t / steps:8.3f} ms/step {c / steps:6.1f} x {a:9.1f} us  {n}")


if __name__ == "__main__":
    main()
