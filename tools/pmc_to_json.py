#!/usr/bin/env python3
"""Summarises gpurun_out/prof_<tag>_<size>/ (tools/profile_round.sh <tag> <size>) into profiles/:
  profiles/<tag>_kernel_stats[_<size>].csv      rocprofv3 --kernel-trace --stats summary of bench.py
  profiles/<tag>_pmc_summary[_<size>].json      per-kernel averages of every collected counter
  profiles/pmc_traffic.json                     HBM bytes and VALU instructions per launch of each kernel
                                                class and workload, and the clock of the counter pass;
                                                read by bench.py (entries of other sizes are kept)

HBM bytes follow the MI355X counter notes: FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950
FETCH_SIZE counts 128-B requests as 64 B for wide coalesced streaming reads, so the read side is
doubled (that correction is calibrated for 16 B/lane streams; the gather kernels' dword loads are
reported with the same factor and flagged "uncalibrated").
"""
import collections
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
size = int(sys.argv[2]) if len(sys.argv) > 2 else 512
suffix = "" if size == 512 else "_%d" % size
workload = "%d^3" % size
src = os.path.join(ROOT, "gpurun_out", "prof_%s_%d" % (tag, size))
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)


def kclass(name):
    for key, cls in (("k_jacobi_tb", "jacobi"), ("k_jacobi_deep", "jacobi"), ("k_jacobi_1", "jacobi_single"), ("k_advect_velocity_lds", "advect_velocity"),
                     ("k_advect_component_global", "advect_velocity_tail"), ("k_advect_scalars", "advect_scalars"),
                     ("k_divergence", "divergence"), ("k_gradient", "gradient"), ("k_force", "force"), ("k_source", "source")):
        if key in name:
            return cls
    return None


stats = os.path.join(src, "stats")
for dirpath, _, files in os.walk(stats):
    for f in files:
        if f.endswith("kernel_stats.csv"):
            shutil.copy(os.path.join(dirpath, f), os.path.join(dst, tag + "_kernel_stats" + suffix + ".csv"))

# the timing at creation launches every tile variant of the blocked Jacobi kernel a few times: only
# the variant the steps themselves use (the one with the most dispatches) is summarised
variant_count = collections.Counter()
for dirpath, _, files in os.walk(src):
    for f in files:
        if f.endswith("counter_collection.csv"):
            for row in csv.DictReader(open(os.path.join(dirpath, f))):
                if "k_jacobi_tb<0" in row["Kernel_Name"] or "k_jacobi_deep" in row["Kernel_Name"]:
                    variant_count[row["Kernel_Name"]] += 1
jacobi_variant = variant_count.most_common(1)[0][0] if variant_count else None

summary = collections.defaultdict(lambda: collections.defaultdict(list))
for sub in sorted(os.listdir(src)):
    path = os.path.join(src, sub)
    if not (os.path.isdir(path) and sub.startswith("pmc_")):
        continue
    for dirpath, _, files in os.walk(path):
        for f in files:
            if not f.endswith("counter_collection.csv"):
                continue
            for row in csv.DictReader(open(os.path.join(dirpath, f))):
                cls = kclass(row["Kernel_Name"])
                if cls == "jacobi" and row["Kernel_Name"] != jacobi_variant:
                    continue
                if cls:
                    summary[cls][row["Counter_Name"]].append(float(row["Counter_Value"]))
                    if sub == "pmc_l2":
                        summary[cls]["duration_ns"].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))

out = {k: {c: sum(v) / len(v) for c, v in d.items()} | {"dispatches": len(next(iter(d.values())))} for k, d in summary.items()}
# the clock the chip held in the counter pass: GRBM_GUI_ACTIVE sums the 8 XCDs
for d in out.values():
    if d.get("GRBM_GUI_ACTIVE") and d.get("duration_ns"):
        d["clock_ghz"] = d["GRBM_GUI_ACTIVE"] / 8.0 / d["duration_ns"]
json.dump(out, open(os.path.join(dst, tag + "_pmc_summary" + suffix + ".json"), "w"), indent=1, sort_keys=True)

sys.path.insert(0, ROOT)
from bench import kernel_source_hash  # noqa: E402

path = os.path.join(dst, "pmc_traffic.json")
h = kernel_source_hash()
traffic, kern = {}, {}
try:
    old = json.load(open(path))
    if old.get("source_hash") == h:      # entries of other workloads measured on the same sources stay
        traffic = old.get("kernels", {})
        kern = old.get("jacobi_kernel", {}) if isinstance(old.get("jacobi_kernel"), dict) else {}
except Exception:
    pass
kern[workload] = jacobi_variant
for cls, d in out.items():
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        rd, wr = d["FETCH_SIZE"] * 1024.0 * 2.0, d["WRITE_SIZE"] * 1024.0
        traffic.setdefault(cls, {})[workload] = {
            "hbm_bytes_per_launch": rd + wr, "read_bytes_corrected": rd, "write_bytes": wr,
            "fetch_size_kib_raw": d["FETCH_SIZE"], "write_size_kib_raw": d["WRITE_SIZE"], "read_factor": 2.0,
            "note": "x2 read correction calibrated for 16 B/lane streams" +
                    ("" if cls in ("jacobi",) else "; uncalibrated for this access pattern"),
            "SQ_INSTS_VALU": d.get("SQ_INSTS_VALU"), "clock_ghz": d.get("clock_ghz"),
            "duration_ns_in_counter_pass": d.get("duration_ns"), "source": tag}
# the counter passes belong to the kernel sources they ran on: bench.py reports them only while
# the sources in the tree still hash to this value
json.dump({"source_hash": h, "tag": tag, "jacobi_kernel": kern, "kernels": traffic}, open(path, "w"), indent=1,
          sort_keys=True)
print(json.dumps(traffic.get("jacobi", {}).get(workload, {}), indent=1))
