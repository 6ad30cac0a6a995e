#!/usr/bin/env python
"""One training step out of a rocprofv3 --kernel-trace CSV, in launch order: python scripts/trace_step.py <dir> [which]
-> start offset (ms), duration (us), kernel.  Steps are cut at the fused Adam kernel (multi_tensor_apply)."""
import csv
import glob
import sys


def short(name):
    for pre in ("void sgf::(anonymous namespace)::", "sgf::(anonymous namespace)::", "void at::native::", "void "):
        name = name.replace(pre, "")
    return name[:100]


def main():
    d = sys.argv[1]
    which = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))))
    cuts = [i for i, r in enumerate(rows) if "multi_tensor_apply" in r[2]]
    # the last Adam launch of a step: the next launch is not an Adam one
    ends = [i for k, i in enumerate(cuts) if k + 1 == len(cuts) or cuts[k + 1] != i + 1]
    a, b = ends[which] + 1, ends[which + 1] + 1
    t0 = rows[a][0]
    busy = 0
    for s, e, n in rows[a:b]:
        busy += e - s
        print(f"{(s - t0) / 1e6:8.3f} ms {(e - s) / 1e3:9.1f} us  {short(n)}")
    print(f"step: {b - a} launches, wall {(rows[b - 1][1] - t0) / 1e6:.2f} ms, busy {busy / 1e6:.2f} ms")


if __name__ == "__main__":
    main()
