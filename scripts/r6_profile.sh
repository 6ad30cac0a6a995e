#!/bin/bash
# r06 measurement call: default bench line, kernel traces (uniform + community), golden bf16 reports
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/${1:-r6_profile}
cd $R && mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_golden.py -q -x 2>&1 | tail -5 > $O/golden.txt
timeout 700 python bench.py > $O/bench_default.json 2> $O/bench_default.err
for g in uniform community; do
  timeout 500 rocprofv3 --kernel-trace --stats -d $O/bench_$g -o b --output-format csv -- \
    python bench.py --graph $g --steps 5 --warmup 2 --no-cpu-baseline --no-structured > $O/bench_$g.log 2>&1
done
timeout 200 python scripts/trace_step.py > $O/step_uniform.txt 2> $O/step_uniform.err
find $O -name "*.csv" -size +20M -delete
find $O -name "*.db" -delete
cat $O/golden.txt; head -c 1500 $O/bench_default.json; du -sh $O
