#!/usr/bin/env python
"""Average every counter of scripts/pmc_passes.sh's passes over the dispatches of ONE kernel (name substring)."""
import glob
import os
import sys

import pandas as pd

out, pat = sys.argv[1], sys.argv[2]
df = pd.concat([pd.read_csv(f) for f in glob.glob(os.path.join(out, "p*", "*counter_collection.csv"))])
df = df[df.Kernel_Name.str.contains(pat)]
res = df.groupby("Counter_Name").Counter_Value.mean()
t = pd.concat([pd.read_csv(f) for f in glob.glob(os.path.join(out, "trace", "*kernel_trace.csv"))])
t = t[t.Kernel_Name.str.contains(pat)]
res["launch_ms"] = ((t.End_Timestamp - t.Start_Timestamp) / 1e6).mean()
if "FETCH_SIZE" in res:
    res["HBM_read_GB"] = 2 * res["FETCH_SIZE"] * 1024 / 1e9
    res["HBM_write_GB"] = res["WRITE_SIZE"] * 1024 / 1e9
if "TCC_HIT_sum" in res:
    res["L2_hit_rate"] = res["TCC_HIT_sum"] / (res["TCC_HIT_sum"] + res["TCC_MISS_sum"])
if "SQ_WAVE_CYCLES" in res:
    for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA",
              "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM"):
        if k in res:
            res[k + " / WAVE_CYCLES"] = res[k] / res["SQ_WAVE_CYCLES"]
print(res.to_frame("mean per launch").to_markdown(floatfmt=".4g"))
