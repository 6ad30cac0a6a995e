#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/${1:-r6g}
cd $R && mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_gramx.py -x -q -k "gramb2 or head_backward" 2>&1 | tail -6 > $O/gramx.txt
timeout 900 python -m pytest tests/test_gpu_graphed.py tests/test_gpu_launch_patches.py tests/test_gpu_golden.py tests/test_gpu_model.py -x -q 2>&1 | tail -8 > $O/tests.txt
for v in 0 1; do
  SGF_OVERLAP=$v timeout 200 python bench.py --no-structured --no-cpu-baseline --steps 10 --warmup 3 > $O/bench_overlap$v.json 2> $O/bench_overlap$v.err
  SGF_OVERLAP=$v timeout 200 python bench.py --graph community --no-structured --no-cpu-baseline --steps 10 --warmup 3 > $O/bench_comm_overlap$v.json 2> $O/bench_comm_overlap$v.err
done
SGF_OVERLAP=1 timeout 1200 python -m pytest tests/test_gpu_model.py tests/test_gpu_golden.py tests/test_gpu_blocked.py tests/test_gpu_scale.py -x -q 2>&1 | tail -6 > $O/tests_overlap.txt
cat $O/gramx.txt $O/tests.txt $O/tests_overlap.txt
for f in bench_overlap0 bench_overlap1 bench_comm_overlap0 bench_comm_overlap1; do python -c "
import json,sys
try:
    d=json.loads(open('$O/$f.json').read().strip().splitlines()[-1]); print('$f', round(d['ms_per_step'],3), d['loss'], d['roofline']['mean_launch_ms'])
except Exception as e: print('$f', 'ERR', e)
"; done
