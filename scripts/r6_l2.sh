#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/l2
timeout 900 python scripts/tile_l2_probe.py --out gpurun_out/l2/times.jsonl 2>&1 | grep -v amdgpu.ids | tail -40
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $c | cut -d' ' -f1)
  (cd $R && timeout 600 rocprofv3 --pmc $c -d gpurun_out/l2/$tag -o p --output-format csv -- python scripts/tile_l2_probe.py --once) > $R/gpurun_out/l2/$tag.log 2>&1
done
cd $R
python - <<'PY'
import csv, glob, collections
out = collections.OrderedDict()
for tag in ("FETCH_SIZE", "TCC_HIT_sum"):
    for f in glob.glob(f"gpurun_out/l2/{tag}/**/*counter_collection.csv", recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if "k_spmm_tile" in r["Kernel_Name"]]
        by = collections.OrderedDict()
        for r in rows:
            by.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
        for i, (k, v) in enumerate(sorted(by.items())):
            out.setdefault(i, {}).update(v)
            out[i]["k"] = [r["Kernel_Name"] for r in rows if int(r["Dispatch_Id"]) == k][0][47:65]
for i, v in out.items():
    print(i, {k: (x if isinstance(x, str) else f"{x:.4g}") for k, x in v.items()})
PY
