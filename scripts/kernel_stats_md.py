#!/usr/bin/env python
"""rocprofv3 --kernel-trace --stats output of one bench.py run -> a tracked markdown table under profiles/.

    python scripts/kernel_stats_md.py <trace dir> <bench log> <out.md> <steps in trace> "<command line>"

`steps in trace`: bench.py --steps 5 --warmup 2 runs 2 + 5 steps, then 1 + 5 with each of the two other loss forms = 19
(--mode minibatch: epochs x batches)."""
import csv
import glob
import os
import sys


def short(name):
    for pre in ("void sgf::(anonymous namespace)::", "sgf::(anonymous namespace)::", "void at::native::", "void "):
        name = name.replace(pre, "")
    return name


def main():
    src, log, out, steps, cmd = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), sys.argv[5]
    f = glob.glob(os.path.join(src, "**", "*kernel_stats.csv"), recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    total = sum(float(r["TotalDurationNs"]) for r in rows)
    line = [ln for ln in open(log) if ln.startswith("{")]
    bench = line[-1].strip() if line else "(bench line not captured)"
    with open(out, "w") as o:
        o.write(f"# rocprofv3 --kernel-trace --stats of `{cmd}` (MI355X)\n\n")
        o.write(f"{steps} steps are in the trace plus the one-off graph preparation; `ms_per_step` = total / {steps}; all kernels "
                f"together: {total / 1e6 / steps:.2f} ms per step.\n\nbench line of the profiled run: `{bench[:1200]} ...`\n\n")
        o.write("| kernel | calls | avg_ms | ms_per_step | % |\n|---|---|---|---|---|\n")
        for r in rows[:40]:
            o.write(f"| {short(r['Name'])[:100]} | {int(r['Calls'])} | {float(r['AverageNs']) / 1e6:.4f} | "
                    f"{float(r['TotalDurationNs']) / 1e6 / steps:.3f} | {100 * float(r['TotalDurationNs']) / total:.2f} |\n")


if __name__ == "__main__":
    main()
