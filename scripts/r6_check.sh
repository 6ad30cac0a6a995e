#!/bin/bash
# r06 working call: A/B probe of the node reductions, the whole GPU suite, the default bench line
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/${1:-r6c}
cd $R && mkdir -p $O
timeout 300 python scripts/gramx_probe.py > $O/probe.jsonl 2> $O/probe.err
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/gpu_suite.txt
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err
cat $O/probe.jsonl; cat $O/gpu_suite.txt; head -c 3000 $O/bench_default.json; tail -3 $O/bench_default.err
