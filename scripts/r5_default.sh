#!/bin/bash
# r05: the driver's command (`python bench.py`), timed by the shell, with its legs printed
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r5_last; mkdir -p $O
t0=$(date +%s)
timeout 800 python bench.py > $O/bench_default2.json 2> $O/bench_default2.err
echo "bench.py wall: $(( $(date +%s) - t0 )) s"
python - <<'PY'
import json
j = json.loads(open("gpurun_out/r5_last/bench_default2.json").read().strip().splitlines()[-1])
print(j["value"], j["ms_per_step"], j["roofline"]["mean_launch_ms"])
s = j["structured"]; print(s["ms_per_step"], s["powerlaw"]["ms_per_step"], s["rmat"]["ms_per_step"])
m = s["minibatch_epoch"]; print(m["value"], m["ms_per_epoch"], m["minibatch"])
PY
tail -3 $O/bench_default2.err
