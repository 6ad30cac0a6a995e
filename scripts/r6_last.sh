#!/bin/bash
# the round's closing call: smoke, the driver's line (with roofline.traffic from profiles/r06_spmm_pmc.json), overlap A/B
R=${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r6_last
cd $R && mkdir -p $O
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
s=$(date +%s); timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "default bench wall: $(( $(date +%s) - s )) s" | tee $O/bench_default_wall.txt
[ -n "$SKIP_AB" ] && exit 0
for i in 1 2; do for v in 0 1; do
  SGF_OVERLAP=$v timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-structured 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('SGF_OVERLAP=$v', round(d['ms_per_step'],3), d['roofline']['achieved'], d['roofline'].get('traffic'))" | tee -a $O/overlap_ab.txt
done; done
