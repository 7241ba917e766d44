#!/usr/bin/env python3
"""From a rocprofv3 --hip-trace --kernel-trace run of bench.py (profiles/run_hip_trace.sh): the HIP
calls the host makes between queueing a frame's first stage (k_wave<.,0,.>) and queueing its
resolve (k_resolve), frame by frame, and how many of them wait for the device or copy back to the
host.

    python tools/hip_trace_frames.py gpurun_out/prof_<tag> > profiles/<round>/hip_trace_frames.txt
"""
import collections
import csv
import glob
import os
import sys

src = sys.argv[1]
api = list(csv.DictReader(open(glob.glob(os.path.join(src, "trace", "**", "*hip_api_trace.csv"), recursive=True)[0])))
ker = list(csv.DictReader(open(glob.glob(os.path.join(src, "trace", "**", "*kernel_trace.csv"), recursive=True)[0])))
kname = {r["Correlation_Id"]: r["Kernel_Name"].replace("void ", "").split("(")[0] for r in ker}
api = [r for r in api if not r["Function"].startswith("__hip")]
api.sort(key=lambda r: int(r["Start_Timestamp"]))
WAITS = ("hipStreamSynchronize", "hipDeviceSynchronize", "hipEventSynchronize", "hipMemcpy", "hipMemcpyAsync", "hipMemcpyDtoH", "hipStreamQuery")
frames, cur = [], None
for r in api:
    k = kname.get(r["Correlation_Id"], "") if r["Function"] == "hipLaunchKernel" else ""
    if k.startswith("k_wave<false, 0") or k.startswith("k_wave<true, 0"):
        cur = []
    if cur is not None:
        cur.append(r["Function"] + (f" {k}" if k else ""))
        if k.startswith("k_resolve"):
            frames.append(cur)
            cur = None
print(f"{len(frames)} frames (first stage ... resolve) in the trace")
for i, f in enumerate(frames):
    c = collections.Counter(x.split(" ")[0] for x in f)
    waits = sum(c[w] for w in WAITS)
    launches = collections.Counter(x.split(" ", 1)[1] for x in f if " " in x)
    print(f"frame {i}: {len(f)} HIP calls between the two launches; calls that wait or copy back: {waits}"
          + (f" ({', '.join(f'{w} x{c[w]}' for w in WAITS if c[w])})" if waits else ""))
    if i in (0, len(frames) - 1):
        print("   launches: " + ", ".join(f"{k} x{v}" for k, v in launches.items()))
        print("   other calls: " + ", ".join(f"{k} x{v}" for k, v in c.items() if k != "hipLaunchKernel"))
