#!/bin/bash
# usage: scripts/pmc_passes.sh <outdir> <target command...>   — one rocprofv3 pass per counter group
# (counter passes carry no tracing flags: the pool's gpurun refuses --pmc together with trace domains)
out=$1; shift
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/$out
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1))
  (cd $R && timeout 300 rocprofv3 --pmc $grp -d $out/p$i -o p$i --output-format csv -- "$@") > $R/$out/p$i.log 2>&1
done
(cd $R && timeout 300 rocprofv3 --kernel-trace --stats -d $out/trace -o t --output-format csv -- "$@") > $R/$out/trace.log 2>&1
