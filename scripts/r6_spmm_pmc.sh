#!/bin/bash
# r06: PMC passes of every SpMM kernel on four graphs -> gpurun_out/r6_pmc/spmm_pmc.json (bench.py's roofline.traffic table)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r6_pmc
cd $R && mkdir -p $O && rm -f $O/spmm_pmc.json
for g in uniform community powerlaw rmat; do
  bash scripts/pmc_passes.sh $O/pmc_$g python scripts/spmm_pmc_target.py --graph $g
  cd $R
  python scripts/pmc_summarise.py $O/pmc_$g $O/spmm_pmc.json ogbn-products:$g/bf16 > $O/pmc_$g.md 2> $O/pmc_$g.err
done
find $O -name "*.csv" -size +20M -delete
find $O -name "*.db" -delete
du -sh $O; ls $O
