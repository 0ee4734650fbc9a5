#!/usr/bin/env python3
"""Summarises gpurun_out/flow_<tag>/ (tools/profile_flow.sh) into profiles/<tag>_flow_counters.json:
per flow (plume, swirl) and kernel class, the average of every counter over the dispatches of the
timed steps, the dispatch duration in the counter pass and the clock the chip held
(GRBM_GUI_ACTIVE / 8 XCDs / duration)."""
import collections
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
src = os.path.join(ROOT, "gpurun_out", "flow_" + tag)
CLASSES = (("k_advect_velocity_lds", "advect_velocity"), ("k_advect_scalars_lds", "advect_scalars"),
           ("k_jacobi_tb", "jacobi"), ("k_jacobi_deep", "jacobi"), ("k_gradient", "gradient"), ("k_force", "force"))
out = {}
for flow in ("plume", "swirl"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for sub in ("a", "b"):
        for dirpath, _, files in os.walk(os.path.join(src, "%s_%s" % (flow, sub))):
            for f in files:
                if not f.endswith("counter_collection.csv"):
                    continue
                for row in csv.DictReader(open(os.path.join(dirpath, f))):
                    cls = next((c for k, c in CLASSES if k in row["Kernel_Name"]), None)
                    if not cls:
                        continue
                    acc[cls][row["Counter_Name"]].append(float(row["Counter_Value"]))
                    if row["Counter_Name"] == "GRBM_GUI_ACTIVE":
                        acc[cls]["duration_ns"].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    res = {}
    for cls, d in acc.items():
        # the last dispatches are the timed steps on the developed state; the autotune launches of
        # the Jacobi kernel at creation are short and would skew an average over all of them
        e = {c: sum(v[-3:]) / len(v[-3:]) for c, v in d.items()}
        if e.get("GRBM_GUI_ACTIVE") and e.get("duration_ns"):
            e["clock_ghz"] = e["GRBM_GUI_ACTIVE"] / 8.0 / e["duration_ns"]
        if e.get("SQ_LDS_IDX_ACTIVE"):
            e["lds_conflict_share"] = e.get("SQ_LDS_BANK_CONFLICT", 0.0) / e["SQ_LDS_IDX_ACTIVE"]
        res[cls] = e
    out[flow] = res
json.dump(out, open(os.path.join(ROOT, "profiles", tag + "_flow_counters.json"), "w"), indent=1, sort_keys=True)
for cls in ("advect_velocity", "advect_scalars", "jacobi"):
    for flow in ("plume", "swirl"):
        e = out.get(flow, {}).get(cls, {})
        print(cls, flow, {k: (round(v, 3) if v < 1e4 else "%.4g" % v) for k, v in sorted(e.items())})
