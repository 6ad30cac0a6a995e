#!/bin/bash
# tile SpMM: one launch at d = 256 against two column-half launches (L2 fit of the super-community gathers)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/slice
timeout 600 python scripts/tile_slice_probe.py --graph community --out gpurun_out/slice/probe.jsonl 2>&1 | tail -3
timeout 600 python scripts/tile_slice_probe.py --graph powerlaw --out gpurun_out/slice/probe.jsonl 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $c | cut -d' ' -f1)
  (cd $R && timeout 600 rocprofv3 --pmc $c -d gpurun_out/slice/$tag -o p --output-format csv -- python scripts/tile_slice_probe.py --reps 2) > $R/gpurun_out/slice/$tag.log 2>&1
done
cd $R
python - <<'PY'
import csv, glob, collections
for tag in ("FETCH_SIZE", "TCC_HIT_sum"):
    for f in glob.glob(f"gpurun_out/slice/{tag}/**/*counter_collection.csv", recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if "k_spmm_tile" in r["Kernel_Name"]]
        by = collections.OrderedDict()
        for r in rows:
            by.setdefault((r["Dispatch_Id"], r["Kernel_Name"][:60]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
        for k, v in by.items():
            print(tag, k, v)
PY
