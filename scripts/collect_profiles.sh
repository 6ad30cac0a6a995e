#!/bin/bash
# One GPU-box run that regenerates the round's profile artefacts under gpurun_out/<round>_profiles/ :
#   (ROUND=r4 by default: gpurun_out/r4_profiles)
#   kernel-trace statistics of bench.py on both graphs, PMC passes of the SpMM kernels on both graphs,
#   an MFMA / SQ counter pass over one bench run.  Copy the summaries into profiles/ afterwards.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/${ROUND:-r4}_profiles
cd $R && mkdir -p $O
for g in uniform community; do
  timeout 500 rocprofv3 --kernel-trace --stats -d $O/bench_$g -o b --output-format csv -- \
    python bench.py --graph $g --steps 5 --warmup 2 --no-cpu-baseline --no-structured > $O/bench_$g.log 2>&1
done
if [ -z "$SKIP_PMC" ]; then   # SKIP_PMC=1: the SpMM sources did not change since the last passes (bench.py checks the hash)
bash scripts/pmc_passes.sh $O/pmc_community python scripts/spmm_pmc_target.py --graph community
bash scripts/pmc_passes.sh $O/pmc_uniform python scripts/spmm_pmc_target.py --graph uniform
bash scripts/pmc_passes.sh $O/pmc_powerlaw python scripts/spmm_pmc_target.py --graph powerlaw
python scripts/pmc_summarise.py $O/pmc_community $O/spmm_pmc.json ogbn-products:community/bf16 > $O/pmc_community.md 2> $O/pmc_community.err
python scripts/pmc_summarise.py $O/pmc_uniform $O/spmm_pmc.json ogbn-products:uniform/bf16 > $O/pmc_uniform.md 2> $O/pmc_uniform.err
python scripts/pmc_summarise.py $O/pmc_powerlaw $O/spmm_pmc.json ogbn-products:powerlaw/bf16 > $O/pmc_powerlaw.md 2> $O/pmc_powerlaw.err
fi
# MFMA / SQ busy counters over the attention + Gram kernels of one step (own pass, no tracing)
rocprofv3 -L > $O/counters_list.txt 2>&1
for grp in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  timeout 400 rocprofv3 --pmc $grp -d $O/mfma$i -o m --output-format csv -- \
    python bench.py --graph uniform --steps 2 --warmup 1 --no-cpu-baseline --no-structured > $O/mfma$i.log 2>&1
done
find $O -name "*.csv" -size +20M -delete     # keep the artefact set small (per-dispatch counter dumps of bench are large)
du -sh $O
