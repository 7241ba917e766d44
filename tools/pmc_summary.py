#!/usr/bin/env python3
"""Reduces the rocprofv3 output of profiles/run_profile.sh to two small files:

    python tools/pmc_summary.py gpurun_out/prof_<tag> profiles/<round>/<tag>

writes <prefix>_kernel_stats.csv (the --stats table of the kernel trace) and
<prefix>_pmc_summary.json (per kernel and counter: launches, sum and mean per launch; plus HBM
bytes per launch from FETCH_SIZE / WRITE_SIZE, both in KiB, FETCH_SIZE doubled for the gfx950
under-count of wide reads)."""
import csv
import glob
import json
import os
import shutil
import sys

src, prefix = sys.argv[1], sys.argv[2]
os.makedirs(os.path.dirname(prefix) or ".", exist_ok=True)
stats = glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    shutil.copy(stats[0], prefix + "_kernel_stats.csv")

out = {}
found = glob.glob(os.path.join(src, "pmc_*", "**", "*counter_collection.csv"), recursive=True)
if not found:
    sys.exit(f"{src}: no counter files (pmc_*/**/*counter_collection.csv): nothing written")
for f in found:
    per_dispatch = {}
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"].split("(")[0].replace("void ", "").strip()
        key = (name, row["Dispatch_Id"], row["Counter_Name"])
        per_dispatch[key] = per_dispatch.get(key, 0.0) + float(row["Counter_Value"])
    for (name, _, counter), v in per_dispatch.items():
        c = out.setdefault(name, {}).setdefault(counter, {"launches": 0, "sum": 0.0})
        c["launches"] += 1
        c["sum"] += v
for name, counters in out.items():
    for c in counters.values():
        c["per_launch"] = c["sum"] / max(1, c["launches"])
    if "FETCH_SIZE" in counters and "WRITE_SIZE" in counters:
        counters["hbm_bytes_per_launch"] = int(1024 * (2.0 * counters["FETCH_SIZE"]["per_launch"] + counters["WRITE_SIZE"]["per_launch"]))
keep = {k: v for k, v in out.items() if k.startswith("k_")}
# the library the passes ran on: bench.py prints its source digest in its JSON line (the logs keep it)
digests = set()
for log in glob.glob(os.path.join(src, "*.log")):
    for line in open(log, errors="replace"):
        if line.startswith("{") and '"library"' in line:
            try:
                digests.add(json.loads(line)["library"])
            except (ValueError, KeyError):
                pass
if digests:
    keep["_library_digest"] = sorted(digests)[0] if len(digests) == 1 else "mixed: " + ", ".join(sorted(digests))
json.dump(keep, open(prefix + "_pmc_summary.json", "w"), indent=1, sort_keys=True)
for name in sorted(keep):
    print(name)
    if not isinstance(keep[name], dict):
        print("   ", keep[name])
        continue
    for counter, c in sorted(keep[name].items()):
        if isinstance(c, dict):
            print(f"   {counter:26s} launches {c['launches']:4d}  per launch {c['per_launch']:.4g}")
        else:
            print(f"   {counter:26s} {c}")
