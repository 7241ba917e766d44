"""Renders a workload (at most 16 spp) with the -DRT_DEBUG_HIST variant of the library and saves the
histogram it gathers: python tools/experiments/hist.py <workload> [out.npy] (RT_AMD_LIB names the variant)."""
import ctypes as C, os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import raytracer_amd as rt
from raytracer_amd import _ffi, scenes
from bench import WORKLOADS
wl = sys.argv[1]
name, W, H, spp, _ = WORKLOADS[wl]
spp = min(spp, 16)
s = scenes.build(name); s.upload(0)
canvas = torch.zeros((H, W, 3), dtype=torch.float64, device="cuda")
buf = (C.c_ulonglong * (8*64*64))()
_ffi.lib.rt_debug_hist(None, 1)
s.render_device(canvas.data_ptr(), rt.render_params(width=W, height=H, spp=spp, seed=0))
s.wait(); torch.cuda.synchronize()
assert _ffi.lib.rt_debug_hist(buf, 0) == 0
h = np.frombuffer(buf, dtype=np.uint64).reshape(8, 64, 64).astype(np.float64)
out = sys.argv[2] if len(sys.argv) > 2 else f"gpurun_out/hist_{wl}.npy"
os.makedirs(os.path.dirname(out) or ".", exist_ok=True)
np.save(out, h)
print(wl, "records", h.sum())
