#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r5g; mkdir -p $O; rm -f $O/probe_summary.txt
for p in direct after-eager after-eager-step; do
  timeout 120 python scripts/graph_probe.py --mgc $p > $O/probe_mgc_$p.log 2>&1; echo "mgc $p rc=$?" | tee -a $O/probe_summary.txt
done
timeout 500 /opt/rocm/bin/rocgdb -batch -ex run -ex bt --args python scripts/graph_probe.py --mgc after-eager-step > $O/gdb_mgc.log 2>&1
grep -n "^#" $O/gdb_mgc.log | head -40
for p in direct after-eager after-eager-step; do echo "== $p"; grep -v "UserWarning\|run_backward\|amdgpu.ids" $O/probe_mgc_$p.log | head -14; done
