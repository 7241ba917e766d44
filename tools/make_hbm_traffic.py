#!/usr/bin/env python3
"""profiles/hbm_traffic.json (what bench.py reads for roofline.traffic and roofline.valu_issue) from
a PMC summary made by tools/pmc_summary.py:

    python tools/make_hbm_traffic.py profiles/r01/g_pmc_summary.json sphere_floor_1080p_64spp
"""
import json
import os
import sys

src, workload = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(root, "profiles", "hbm_traffic.json")
pmc = json.load(open(src))
out = json.load(open(dst)) if os.path.exists(dst) else {}
out["_doc"] = (
    "Per kernel, means over every launch in rocprofv3 PMC passes of bench.py's default command (separate --pmc FETCH_SIZE, "
    "--pmc WRITE_SIZE and SQ runs, profiles/run_profile.sh): bytes_per_launch = HBM bytes, FETCH_SIZE doubled for the gfx950 "
    "under-count of wide reads, both counters in KiB; valu_wave_insts_per_launch = SQ_INSTS_VALU; valu_lane_use = "
    "SQ_THREAD_CYCLES_VALU / (64 * SQ_ACTIVE_INST_VALU). Source: " + os.path.relpath(os.path.abspath(src), root) + "."
)
w = {}
for name, c in pmc.items():
    if not name.startswith("k_") or "<true" in name or "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
        continue
    w[name.replace(", ", ",")] = {
        "launches_profiled": c["FETCH_SIZE"]["launches"],
        "fetch_kib_per_launch": c["FETCH_SIZE"]["per_launch"],
        "write_kib_per_launch": c["WRITE_SIZE"]["per_launch"],
        "bytes_per_launch": int(1024 * (2.0 * c["FETCH_SIZE"]["per_launch"] + c["WRITE_SIZE"]["per_launch"])),
        "valu_wave_insts_per_launch": c["SQ_INSTS_VALU"]["per_launch"],
        "valu_lane_use": round(c["SQ_THREAD_CYCLES_VALU"]["per_launch"] / (64.0 * c["SQ_ACTIVE_INST_VALU"]["per_launch"]), 4),
    }
out[workload] = w
json.dump(out, open(dst, "w"), indent=1)
for k, v in w.items():
    print(f"{k:18s} {v['bytes_per_launch'] / 1e9:8.3f} GB/launch  {v['valu_wave_insts_per_launch'] / 1e9:7.3f} G VALU instr  lane use {v['valu_lane_use']}")
