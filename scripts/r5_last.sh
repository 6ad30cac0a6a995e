#!/bin/bash
# r05 closing call: the whole GPU suite, smoke(), the default bench line and the mini-batch leg after the last library change.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r5_last; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu -x > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
tail -4 $O/tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 700 $O/bench_default.json | head -c 400; echo
timeout 300 python bench.py --mode minibatch --steps 4 --warmup 1 > $O/bench_minibatch.json 2> $O/bench_minibatch.err
timeout 200 python scripts/minibatch_sections.py --batches 12 > $O/minibatch_sections.json 2> $O/minibatch_sections.err
python - <<'PY'
import json
for f in ("bench_default", "bench_minibatch"):
    try:
        j = json.loads(open(f"gpurun_out/r5_last/{f}.json").read().strip().splitlines()[-1])
        print(f, j["value"], j["ms_per_step"], json.dumps(j.get("minibatch")), json.dumps(j["roofline"])[:300])
    except Exception as e:
        print(f, "no line", e)
PY
cat $O/minibatch_sections.json
