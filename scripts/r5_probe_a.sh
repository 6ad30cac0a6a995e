#!/bin/bash
# r05 GPU call: new parity tests, the mini-batch epoch leg, the R-MAT leg, kernel traces of the fp32 configurations.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r5b
cd $R && mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_golden.py tests/test_gpu_surface.py tests/test_gpu_sampler.py -x -q -m gpu -k "fixture or per_head or sampler" > $O/tests_a.log 2>&1
tail -5 $O/tests_a.log
timeout 600 python -m pytest tests/test_gpu_scale.py -x -q -m gpu -s -k "gradients_at_full_size or trajectory" > $O/tests_b.log 2>&1
tail -5 $O/tests_b.log
timeout 300 python bench.py --mode minibatch --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_minibatch.json 2> $O/bench_minibatch.err
tail -c 1500 $O/bench_minibatch.json
timeout 300 rocprofv3 --kernel-trace --stats -d $O/mb_trace -o t --output-format csv -- python bench.py --mode minibatch --steps 1 --warmup 1 --no-cpu-baseline > $O/mb_trace.log 2>&1
timeout 300 python bench.py --graph rmat --steps 5 --warmup 3 --no-cpu-baseline --no-structured > $O/bench_rmat.json 2> $O/bench_rmat.err
tail -c 900 $O/bench_rmat.json
for w in "ogbn-arxiv f32 0" "pokec f32 0" "ogbn-arxiv f32 recipe"; do
  set -- $w
  timeout 400 rocprofv3 --kernel-trace --stats -d $O/trace_$1_$2_$3 -o b --output-format csv -- \
    python bench.py --workload $1 --dtype $2 --dropout $3 --steps 5 --warmup 2 --no-cpu-baseline --no-structured > $O/bench_$1_$2_$3.log 2>&1
  tail -c 300 $O/bench_$1_$2_$3.log
done
find $O -name "*.csv" -size +30M -delete
du -sh $O
