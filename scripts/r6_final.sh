#!/bin/bash
# r06 final measurement call: bench lines of every BASELINE single-GPU configuration, kernel traces, MFMA counters
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/${1:-r6_final}
cd $R && mkdir -p $O
timeout 700 python bench.py > $O/bench_default.json 2> $O/bench_default.err
for w in "pokec bf16 0" "pokec f32 0" "ogbn-arxiv f32 0" "ogbn-arxiv f32 recipe" "cora f32 recipe" "cora f32 0" "papers100M-shard8 bf16 0"; do
  set -- $w
  extra="--no-structured"; [ "$1" = "cora" ] && extra=""
  timeout 400 python bench.py --workload $1 --dtype $2 --dropout $3 --steps 10 --warmup 3 $extra $( [ "$1" = "cora" ] || echo --no-cpu-baseline ) > $O/bench_$1_$2_$3.json 2> $O/bench_$1_$2_$3.err
done
timeout 300 python bench.py --mode minibatch --steps 4 --warmup 2 --no-cpu-baseline > $O/bench_minibatch.json 2> $O/bench_minibatch.err
SGF_PREP_STREAM=0 timeout 300 python bench.py --mode minibatch --steps 4 --warmup 2 --no-cpu-baseline > $O/bench_minibatch_noprep.json 2> $O/bench_minibatch_noprep.err
timeout 200 python bench.py --mode minibatch --workload pokec --steps 4 --warmup 2 --no-cpu-baseline > $O/bench_minibatch_pokec.json 2> $O/bench_minibatch_pokec.err
# per-kernel durations with ONE stream (a kernel's own time), the step timeline with the two branches overlapped
for g in uniform community; do
  SGF_OVERLAP=0 timeout 500 rocprofv3 --kernel-trace --stats -d $O/bench_$g -o b --output-format csv -- \
    python bench.py --graph $g --steps 5 --warmup 2 --no-cpu-baseline --no-structured > $O/bench_$g.log 2>&1
done
timeout 500 rocprofv3 --kernel-trace --stats -d $O/bench_uniform_overlap -o b --output-format csv -- \
    python bench.py --graph uniform --steps 5 --warmup 2 --no-cpu-baseline --no-structured > $O/bench_uniform_overlap.log 2>&1
python scripts/trace_step.py $O/bench_uniform 3 > $O/step_uniform_one_stream.txt 2>> $O/trace.err
python scripts/trace_step.py $O/bench_uniform_overlap 3 > $O/step_uniform_overlap.txt 2>> $O/trace.err
python scripts/trace_step.py $O/bench_community 3 > $O/step_community_one_stream.txt 2>> $O/trace.err
i=0
for grp in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  SGF_OVERLAP=0 timeout 400 rocprofv3 --pmc $grp -d $O/mfma$i -o m --output-format csv -- \
    python bench.py --graph uniform --steps 2 --warmup 1 --no-cpu-baseline --no-structured > $O/mfma$i.log 2>&1
done
# HBM bytes of the new reduction kernels (separate passes, never combined with tracing)
for c in FETCH_SIZE WRITE_SIZE; do
  SGF_OVERLAP=0 timeout 400 rocprofv3 --pmc $c -d $O/pmc_$c -o p --output-format csv -- \
    python bench.py --graph uniform --steps 2 --warmup 1 --no-cpu-baseline --no-structured > $O/pmc_$c.log 2>&1
done
find $O -name "*kernel_trace.csv" -size +30M -delete
find $O -name "*.db" -delete
du -sh $O; ls $O | head -60
