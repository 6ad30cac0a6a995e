#!/bin/bash
# r05 final measurement call: the round's bench lines + profile artefacts under gpurun_out/r5_final (copied to profiles/ afterwards)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r5_final
cd $R && mkdir -p $O
if [ -z "$SKIP_BENCH" ]; then
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err
for w in "pokec bf16 0" "pokec f32 0" "ogbn-arxiv f32 0" "ogbn-arxiv f32 recipe" "cora f32 recipe" "cora f32 0" "papers100M-shard8 bf16 0"; do
  set -- $w
  extra="--no-structured"; [ "$1" = "cora" ] && extra=""
  timeout 400 python bench.py --workload $1 --dtype $2 --dropout $3 --steps 10 --warmup 3 $extra $( [ "$1" = "cora" ] || echo --no-cpu-baseline ) > $O/bench_$1_$2_$3.json 2> $O/bench_$1_$2_$3.err
done
timeout 300 python bench.py --mode minibatch --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_minibatch.json 2> $O/bench_minibatch.err
timeout 200 python scripts/minibatch_sections.py > $O/minibatch_sections.json 2> $O/minibatch_sections.err
fi
if [ -z "$SKIP_TRACE" ]; then
for g in uniform community; do
  timeout 500 rocprofv3 --kernel-trace --stats -d $O/bench_$g -o b --output-format csv -- \
    python bench.py --graph $g --steps 5 --warmup 2 --no-cpu-baseline --no-structured > $O/bench_$g.log 2>&1
done
fi
if [ -z "$SKIP_PMC" ]; then
for g in uniform community powerlaw rmat; do
  bash scripts/pmc_passes.sh $O/pmc_$g python scripts/spmm_pmc_target.py --graph $g
  python scripts/pmc_summarise.py $O/pmc_$g $O/spmm_pmc.json ogbn-products:$g/bf16 > $O/pmc_$g.md 2> $O/pmc_$g.err
done
fi
if [ -z "$SKIP_MFMA" ]; then
i=0
for grp in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  timeout 400 rocprofv3 --pmc $grp -d $O/mfma$i -o m --output-format csv -- \
    python bench.py --graph uniform --steps 2 --warmup 1 --no-cpu-baseline --no-structured > $O/mfma$i.log 2>&1
done
fi
find $O -name "*.csv" -size +20M -delete
find $O -name "*.db" -delete
du -sh $O
