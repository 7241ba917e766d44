#!/usr/bin/env python3
"""Loops of one kernel of rt_device.hip and what they hold: compiles the device code to assembly
with line tables (as tools/isa_census.py does), assigns every basic block to the innermost loop
the compiler's block comments name for it ("in Loop: Header=..."), and prints per loop the vector,
scalar, vector-memory and LDS instructions of its own body (inner loops taken out), with the source
lines that contribute most, so that a loop can be told from its neighbours. With --trips the
dynamic count per wave is added up: straight-line code once, every loop body times its measured
trip count per wave (rt_launch_info.wave_iters_* / waves).

    python tools/isa_loops.py --kernel k_waveILb0ELi0ELb0E --trips L1=...,L2=...
"""
import argparse
import collections
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from isa_census import CSRC, classify  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kernel", default="k_waveILb0ELi0ELb0E", help="substring of the mangled kernel name")
    ap.add_argument("--asm", default=None)
    ap.add_argument("--trips", default="", help="trip counts per wave by loop number, e.g. 1=3.2,2=15.7")
    a = ap.parse_args()
    asm = a.asm
    if not asm:
        asm = os.path.join(tempfile.gettempdir(), "rt_device_census.s")
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math",
               "-gline-tables-only", "--cuda-device-only", "-S", "-o", asm, "rt_device.hip"]
        subprocess.run(cmd, cwd=CSRC, check=True, stderr=subprocess.DEVNULL)
    files = {}
    insts = []   # (class, file, line, text, loop header name or None)
    depth_of = {}  # loop header name -> depth
    order = []   # loop header names in order of appearance
    in_kernel = False
    cur = ("?", 0)
    block_loop = None
    pending_label = None  # a block label whose annotation may continue on the next comment lines
    for l in open(asm):
        m = re.match(r"\s*\.file\s+(\d+)\s+\"[^\"]*\"\s+\"([^\"]+)\"", l) or re.match(r"\s*\.file\s+(\d+)\s+\"([^\"]+)\"", l)
        if m:
            files[int(m.group(1))] = os.path.basename(m.group(2))
            continue
        if re.match(r"^_Z\w+:", l):
            in_kernel = a.kernel in l
            block_loop = None
            continue
        if not in_kernel:
            continue
        if ".end_amdhsa_kernel" in l or l.strip().startswith(".section"):
            in_kernel = False
            continue
        m = re.match(r"^(?:\.L(BB\d+_\d+)|; %bb\.(\d+)):(.*)$", l)
        if m:  # a basic block starts; the compiler notes the loop it belongs to beside it
            pending_label = m.group(1)
            block_loop = None
            l = m.group(3)
        if pending_label is not None or re.match(r"^\s+;", l):
            h = re.search(r"in Loop: Header=(BB\d+_\d+) Depth=(\d+)", l)
            if h:
                block_loop = h.group(1)
            h = re.search(r"This (?:Inner )?Loop Header: Depth=(\d+)", l)
            if h and pending_label:
                block_loop = pending_label
                depth_of[block_loop] = int(h.group(1))
                order.append(block_loop)
            if m:
                continue
        m2 = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", l)
        if m2:
            cur = (files.get(int(m2.group(1)), "?"), int(m2.group(2)))
            continue
        t = l.strip()
        if not t or t.startswith((".", ";")) or t.endswith(":"):
            continue
        pending_label = None
        insts.append((classify(t.split()[0]), cur[0], cur[1], t, block_loop))
    loops = [(name, depth_of[name]) for name in order]
    number = {name: n for n, (name, _) in enumerate(loops, 1)}
    owner = [number.get(x[4], 0) for x in insts]
    src = {}
    for name in ("rt_kernels.hip.h", "rt_device.hip"):
        src[name] = open(os.path.join(CSRC, name)).read().split("\n")
    trips = {}
    for kv in filter(None, a.trips.split(",")):
        k, v = kv.split("=")
        trips[int(k)] = float(v)
    cols = ["valu_f64", "valu_trans", "valu_imul", "valu_other", "salu", "vmem", "lds"]
    print(f"kernel *{a.kernel}*: {len(insts)} instructions, {len(loops)} loops")
    print(f"{'':34s}" + "".join(f"{c:>11s}" for c in cols) + f"{'VALU':>7s}{'x trips':>9s}{'= VALU/wave':>13s}")
    total = 0.0
    for n in range(0, len(loops) + 1):
        c = collections.Counter()
        lines = collections.Counter()
        for i, (cls, f, line, t, _) in enumerate(insts):
            if owner[i] == n:
                c[cls] += 1
                if cls.startswith("valu"):
                    lines[(f, line)] += 1
        valu = sum(v for k, v in c.items() if k.startswith("valu"))
        name = "straight-line code" if n == 0 else f"loop {n} ({loops[n - 1][0]}, depth {loops[n - 1][1]})"
        tr = 1.0 if n == 0 else trips.get(n)
        dyn = "" if tr is None else f"{valu * tr:13.0f}"
        if tr is not None:
            total += valu * tr
        print(f"{name:34s}" + "".join(f"{c.get(k, 0):11d}" for k in cols) + f"{valu:7d}" + (f"{tr:9.1f}" if tr is not None else f"{'?':>9s}") + dyn)
        for (f, line), k in lines.most_common(3):
            text = src[f][line - 1].strip()[:90] if f in src and 0 < line <= len(src[f]) else ""
            print(f"      {k:4d}  {f}:{line}  {text}")
    if trips:
        print(f"vector instructions per wave with these trip counts: {total:.0f}")


if __name__ == "__main__":
    main()
