#!/bin/bash
# r05: mini-batch epoch against the host thread count of the trainer's CPU lines (noise from the shared host shows up here),
# then a kernel trace of the epoch (eager launches and hipGraph replays).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r5g; mkdir -p $O
for t in ${THREADS:-2 4 8 2 4 8}; do
  OMP_NUM_THREADS=$t timeout 200 python bench.py --mode minibatch --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/mb_t$t.json
  python - $t <<'PY'
import json, sys
j = json.loads(open(f"gpurun_out/r5g/mb_t{sys.argv[1]}.json").read())
m = j["minibatch"]
print("threads", sys.argv[1], round(j["value"] / 1e6, 2), "M nodes/s", m["per_batch_ms_wall"], "host", m["per_batch_ms_host_issue"], "gpu", m["per_batch_ms_on_the_gpu_timeline"])
PY
done
cat /proc/loadavg
export TMPDIR=/tmp
for mode in 0 1; do
  SGF_BATCH_GRAPH=$mode timeout 400 rocprofv3 --kernel-trace --stats -d $O/mbtrace$mode -o b --output-format csv -- \
    python bench.py --mode minibatch --steps 2 --warmup 1 --no-cpu-baseline > $O/mbtrace$mode.log 2> $O/mbtrace$mode.err
  n=$([ $mode = 0 ] && echo 75 || echo 100)
  python scripts/kernel_stats_md.py $O/mbtrace$mode $O/mbtrace$mode.log $O/mb_kernels_graph$mode.md $n "SGF_BATCH_GRAPH=$mode python bench.py --mode minibatch --steps 2 --warmup 1 --no-cpu-baseline"
  find $O/mbtrace$mode -name "*kernel_trace.csv" -delete
  head -45 $O/mb_kernels_graph$mode.md | cut -c1-160
done
