#!/bin/bash
# usage: scripts/pmc_diag.sh <outdir> <target command...> — diagnostic counter groups (instruction cache, LDS queues,
# vector-memory queues, occupancy) for ONE kernel; one rocprofv3 pass per group, no tracing flags.
out=$1; shift
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/$out
i=0
for grp in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_INSTS_BRANCH" \
           "SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_UNALIGNED_STALL SQ_INST_LEVEL_LDS" \
           "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_INST_CYCLES_VMEM_RD SQ_WAVES" \
           "SQ_BUSY_CU_CYCLES SQ_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_MISC" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VALU"; do
  i=$((i+1))
  (cd $R && timeout 300 rocprofv3 --pmc $grp -d $out/p$i -o p$i --output-format csv -- "$@") > $R/$out/p$i.log 2>&1
done
(cd $R && timeout 300 rocprofv3 --kernel-trace --stats -d $out/trace -o t --output-format csv -- "$@") > $R/$out/trace.log 2>&1
