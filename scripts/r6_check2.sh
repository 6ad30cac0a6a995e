#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/${1:-r6f}
cd $R && mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_gramx.py -x -q -k "gramb2" 2>&1 | tail -8 > $O/gramb2.txt
timeout 300 python scripts/gramx_probe.py --rounds 3 > $O/probe.jsonl 2> $O/probe.err
timeout 900 python -m pytest tests/test_gpu_graphed.py tests/test_gpu_launch_patches.py tests/test_gpu_golden.py tests/test_gpu_model.py -x -q 2>&1 | tail -8 > $O/tests.txt
timeout 600 python -m pytest tests/test_gpu_scale.py -x -q -k "trajectory" 2>&1 | tail -8 > $O/traj.txt
timeout 300 python bench.py --no-structured --no-cpu-baseline --steps 10 --warmup 3 > $O/bench_uniform.json 2> $O/bench_uniform.err
SGF_GRAM_BN2=0 timeout 300 python bench.py --no-structured --no-cpu-baseline --steps 10 --warmup 3 > $O/bench_uniform_nobn2.json 2> $O/bench_uniform_nobn2.err
for v in 0 1; do
  SGF_OVERLAP=$v timeout 200 python bench.py --no-structured --no-cpu-baseline --steps 10 --warmup 3 > $O/bench_overlap$v.json 2> $O/bench_overlap$v.err
  SGF_OVERLAP=$v timeout 200 python bench.py --workload ogbn-arxiv --dtype f32 --no-structured --no-cpu-baseline --steps 20 --warmup 5 > $O/bench_arxiv_overlap$v.json 2> $O/bench_arxiv_overlap$v.err
done
cat $O/gramb2.txt $O/tests.txt $O/traj.txt; cat $O/probe.jsonl; tail -2 $O/probe.err
for f in bench_uniform bench_uniform_nobn2 bench_overlap0 bench_overlap1 bench_arxiv_overlap0 bench_arxiv_overlap1; do python -c "
import json,sys
try:
    d=json.loads(open('$O/$f.json').read().strip().splitlines()[-1]); print('$f', round(d['ms_per_step'],3), d['loss'])
except Exception as e: print('$f', 'ERR', e)
"; done
