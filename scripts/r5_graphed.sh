#!/bin/bash
# r05: the captured mini-batch step — parity tests, the epoch bench with and without replays, the per-section probe.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r5g; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_graphed.py tests/test_gpu_r05.py tests/test_gpu_gemm.py tests/test_gpu_surface.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
tail -4 $O/tests.log
for i in 1 2; do
timeout 300 python bench.py --mode minibatch --steps 4 --warmup 1 $([ $i = 2 ] && echo --no-cpu-baseline) > $O/mb_graph$i.json 2> $O/mb_graph$i.err
done
SGF_BATCH_GRAPH=0 timeout 300 python bench.py --mode minibatch --steps 4 --warmup 1 --no-cpu-baseline > $O/mb_eager.json 2> $O/mb_eager.err
timeout 200 python scripts/minibatch_sections.py --batches 12 > $O/sections_graph.json 2> $O/sections_graph.err
python - <<'PY'
import json
for f in ("mb_graph1", "mb_graph2", "mb_eager"):
    try:
        j = json.loads(open(f"gpurun_out/r5g/{f}.json").read().strip().splitlines()[-1])
        print(f, j["value"], j["ms_per_step"], json.dumps(j["minibatch"]), json.dumps(j["roofline"]))
    except Exception as e:
        print(f, "no line", e)
PY
tail -2 $O/mb_graph1.err; tail -c 1500 $O/sections_graph.json
