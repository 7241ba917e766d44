#!/usr/bin/env python3
"""profiles/hbm_traffic.json (what bench.py reads for roofline.traffic and roofline.valu_issue) from
a PMC summary made by tools/pmc_summary.py:

    python tools/make_hbm_traffic.py profiles/r02/workloads/sphere_floor_1080p_64spp_pmc_summary.json sphere_floor_1080p_64spp [revision]

The launches of one stage are one entry: the kernel's third template argument (the variant that
loops over the blocks of work left over by a launch sized from the previous frame, queued behind
the plain one as part of the same logged launch) is folded away — its counts are added, the
number of launches is the plain variant's. `revision` (default: git's HEAD) is recorded as
_library_revision, the source state the library that was profiled was built from.
"""
import json
import os
import re
import subprocess
import sys

src, workload = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(root, "profiles", "hbm_traffic.json")
pmc = json.load(open(src))
out = json.load(open(dst)) if os.path.exists(dst) else {}
out["_doc"] = (
    "Per workload (bench.py --workload) and kernel, means over every launch in rocprofv3 PMC passes of bench.py (separate --pmc FETCH_SIZE, "
    "--pmc WRITE_SIZE and SQ runs, profiles/run_profile.sh): bytes_per_launch = HBM bytes, FETCH_SIZE doubled for the gfx950 "
    "under-count of wide reads, both counters in KiB; valu_wave_insts_per_launch = SQ_INSTS_VALU; valu_lane_use = "
    "SQ_THREAD_CYCLES_VALU / (64 * SQ_ACTIVE_INST_VALU); tcc_requests_per_launch = TCC_HIT_sum + TCC_MISS_sum (requests the L2s received), "
    "tcp_accesses_per_launch = TCP_TOTAL_CACHE_ACCESSES_sum (accesses of the CUs' vector L1s), valu_insts_per_wave = SQ_INSTS_VALU / SQ_WAVES. Launches of one stage (k_wave<false,S,*>) are one entry. _sources names the "
    "summary each workload was taken from, _library_digests the source digest (rt_source_digest) of the library each was taken on, "
    "_library_revision git's HEAD when this file was last written."
)
out.setdefault("_sources", {})[workload] = os.path.relpath(os.path.abspath(src), root)
out.setdefault("_library_digests", {})[workload] = pmc.get("_library_digest", "unstamped")  # rt_source_digest() of the profiled library
if len(sys.argv) > 3:
    out["_library_revision"] = sys.argv[3]
else:
    out["_library_revision"] = subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=root, capture_output=True, text=True).stdout.strip()
need = ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_ACTIVE_INST_VALU")
extra = ("TCC_HIT_sum", "TCC_MISS_sum", "TCP_TOTAL_CACHE_ACCESSES_sum", "SQ_WAVES", "SQ_INSTS_VMEM_RD")  # taken along when the passes hold them
merged = {}
for name, c in pmc.items():
    if not isinstance(c, dict) or not name.startswith("k_") or "<true" in name or any(k not in c for k in need):
        continue
    # k_wave<COUNT, STAGE, LOOP, QUANT, EMIT, DEEP>: the launches of one stage are one entry, whichever form they run in
    flat_name = name.replace(" ", "")
    mw = re.match(r"^(k_wave<false,\d+),(false|true)(?:,(?:false|true))*>$", flat_name)
    key = mw.group(1) + ">" if mw else flat_name
    # the looping variant is queued behind the plain one as part of the same logged launch: its
    # counts are added, the number of launches is the plain variant's
    looping = bool(mw) and mw.group(2) == "true"
    m = merged.setdefault(key, {k: {"launches": 0, "sum": 0.0} for k in need + extra})
    for k in need + extra:
        if k not in c:
            continue
        if not looping:
            m[k]["launches"] += c[k]["launches"]
        m[k]["sum"] += c[k]["sum"]
w = {}
for key, c in merged.items():
    for k in need + extra:
        c[k]["per_launch"] = c[k]["sum"] / max(1, c[k]["launches"])
    w[key] = {
        "launches_profiled": c["FETCH_SIZE"]["launches"],
        "fetch_kib_per_launch": c["FETCH_SIZE"]["per_launch"],
        "write_kib_per_launch": c["WRITE_SIZE"]["per_launch"],
        "bytes_per_launch": int(1024 * (2.0 * c["FETCH_SIZE"]["per_launch"] + c["WRITE_SIZE"]["per_launch"])),
        "valu_wave_insts_per_launch": c["SQ_INSTS_VALU"]["per_launch"],
        "valu_lane_use": round(c["SQ_THREAD_CYCLES_VALU"]["per_launch"] / (64.0 * c["SQ_ACTIVE_INST_VALU"]["per_launch"]), 4),
    }
    if c["TCC_HIT_sum"]["launches"] and c["TCC_MISS_sum"]["launches"]:
        req = c["TCC_HIT_sum"]["per_launch"] + c["TCC_MISS_sum"]["per_launch"]
        w[key]["tcc_requests_per_launch"] = req
        w[key]["tcc_hit_rate"] = round(c["TCC_HIT_sum"]["per_launch"] / max(1.0, req), 4)
    if c["TCP_TOTAL_CACHE_ACCESSES_sum"]["launches"]:
        w[key]["tcp_accesses_per_launch"] = c["TCP_TOTAL_CACHE_ACCESSES_sum"]["per_launch"]
    if c["SQ_WAVES"]["launches"]:
        w[key]["waves_per_launch"] = c["SQ_WAVES"]["per_launch"]
        w[key]["valu_insts_per_wave"] = round(c["SQ_INSTS_VALU"]["per_launch"] / max(1.0, c["SQ_WAVES"]["per_launch"]), 1)
out[workload] = w
json.dump(out, open(dst, "w"), indent=1)
for k, v in w.items():
    print(f"{k:18s} {v['bytes_per_launch'] / 1e9:8.3f} GB/launch  {v['valu_wave_insts_per_launch'] / 1e9:7.3f} G VALU instr  lane use {v['valu_lane_use']}")
