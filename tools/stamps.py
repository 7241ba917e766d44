#!/usr/bin/env python3
"""Where a wave's lifetime goes, by code section and kernel stage: renders frames of a workload
with the diagnostic build of the library (-DRT_STAMPS, tools/experiments/build_variant.sh stamps
-DRT_STAMPS; RT_AMD_LIB names it) and prints the cycle shares its kernels accumulated.

    RT_AMD_LIB=exp_build/librt_stamps.so python tools/stamps.py --workload spot_4k_256spp
"""
import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import raytracer_amd as rt
from raytracer_amd import _ffi, scenes

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import WORKLOADS  # noqa: E402

SECTIONS = ["load + normal + texture", "rejection loops", "rest of shading + ray set-up", "-", "node steps", "triangle tests", "spheres",
            "sky, sample, queue append"]
ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="spot_4k_256spp", choices=sorted(WORKLOADS))
ap.add_argument("--frames", type=int, default=2)
a = ap.parse_args()
if not hasattr(_ffi.lib, "rt_debug_stamps"):
    raise SystemExit("the library at hand is not the diagnostic build (RT_AMD_LIB)")
name, W, H, spp, _ = WORKLOADS[a.workload]
s = scenes.build(name)
s.upload(0)
canvas = torch.zeros((H, W, 3), dtype=torch.float64, device="cuda")
buf = (C.c_ulonglong * 24)()
for f in range(a.frames + 1):
    if f == 1:
        s.wait()
        _ffi.lib.rt_debug_stamps(None, 1)
    s.render_device(canvas.data_ptr(), rt.render_params(width=W, height=H, spp=spp, seed=0, first_sample=f * spp))
s.wait()
torch.cuda.synchronize()
assert _ffi.lib.rt_debug_stamps(buf, 0) == 0
for stage, label in enumerate(["first stage", "launches over the queue", "tail"]):
    v = [buf[stage * 8 + i] for i in range(8)]
    tot = sum(v) or 1
    print(f"{a.workload} {label}: {tot / a.frames / 1e9:.3f} G wave-cycles per frame")
    for i, x in enumerate(v):
        if x:
            print(f"    {SECTIONS[i]:32s} {100.0 * x / tot:5.1f} %")
