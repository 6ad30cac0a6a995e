#!/usr/bin/env python
"""tests/golden/bf16_bounds.json from the errors the bf16 production-fixture test measured on an MI355X
(gpurun_out/golden_bf16_reports.jsonl, written by tests/test_gpu_golden.py): every bound = 2 x the measured value (VERDICT r05
item 4), with a floor of 5e-5 on the normalised gradient errors (the attention weights sit at 1e-5, where a different but
equally valid summation order moves the value by more than a factor of two).

    python scripts/make_bf16_bounds.py gpurun_out/golden_bf16_reports.jsonl
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rows = {}
for ln in open(sys.argv[1]):
    r = json.loads(ln)
    rows[r.pop("fixture")] = r          # the last report of a fixture wins
out = {"_what": "per-fixture, per-tensor bounds of tests/test_gpu_golden.py::test_bf16_module_matches_reference_fixture_at_production_width "
                "= 2 x the error measured on MI355X in round 6 (floor 5e-5 on gradients); regenerate with scripts/make_bf16_bounds.py",
       "_measured": rows}
for name, r in rows.items():
    b = {}
    for k, v in r.items():
        if k == "logits_scale":
            continue
        b[k] = max(2.0 * v, 5e-5) if k.startswith("grad/") else 2.0 * v
    b["loss_err"] = max(b["loss_err"], 1e-4)
    out[name] = b
path = os.path.join(ROOT, "tests", "golden", "bf16_bounds.json")
json.dump(out, open(path, "w"), indent=1, sort_keys=True)
print(path, {k: len(v) for k, v in out.items() if not k.startswith("_")})
