#!/usr/bin/env python3
"""Instruction census of one kernel of rt_device.hip: compiles the device code to assembly with
line tables, attributes every instruction of the kernel to the source function it was inlined
from (and, inside the big functions, to a named part of it), and prints per part how many vector
(split into binary64 / transcendental / other), scalar, vector-memory and LDS instructions the
code holds. Static counts; multiply a loop body's count by its measured trip count per wave
(rt_launch_info.wave_iters_*) for the dynamic number.

    python tools/isa_census.py                       # k_wave<false,0>
    python tools/isa_census.py --kernel k_waveILb0ELi1E --trips inner=24.8,leaf=4.0,reject=6.1
"""
import argparse
import bisect
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "raytracer_amd", "csrc")

# parts of the big functions: (file, function, marker regex that starts the part, part name);
# a part runs from its marker to the next marker of the same function (or the function's end)
PARTS = [
    ("rt_kernels.hip.h", "trav_run", r"while \(cur != RT_REF_SENTINEL\)", "trav_run: outer loop"),
    ("rt_kernels.hip.h", "trav_run", r"while \(cur >= 0\)", "trav_run: node step (inner loop)"),
    ("rt_kernels.hip.h", "trav_run", r"if \(cur == RT_REF_SENTINEL\) break;", "trav_run: leaf loop (triangle test)"),
    ("rt_kernels.hip.h", "trav_run", r"^\s*cur = \*sp;\s*$", "trav_run: pop after leaf"),
    ("rt_kernels.hip.h", "shade_hit", r"Rng rng;", "shade_hit: material set-up, normal map, texel"),
    ("rt_kernels.hip.h", "shade_hit", r"int phase = 0", "shade_hit: rejection loop"),
    ("rt_kernels.hip.h", "shade_hit", r"diffuse_out = diffuse \* diffuse_out \+ normal", "shade_hit: scatter direction"),
    ("rt_kernels.hip.h", "shade_hit", r"// Dielectric", "shade_hit: dielectric"),
]


def function_starts(path):
    """[(line, name)] of the function definitions of a source file (top-level or in a namespace)."""
    out = []
    rx = re.compile(r"^\s*(?:template\s*<[^>]*>\s*)?(?:static\s+|extern\s+\"C\"\s+)?(?:__host__\s+)?(?:__device__|__global__|static|inline|extern)[^;=]*?\b([A-Za-z_]\w*)\s*\([^;]*$")
    lines = open(path).read().split("\n")
    for i, l in enumerate(lines, 1):
        if l.startswith("  ") and not l.lstrip().startswith("__device__"):
            continue  # bodies
        m = rx.match(l)
        if m and m.group(1) not in ("if", "for", "while", "switch", "return", "__launch_bounds__", "defined"):
            out.append((i, m.group(1)))
        elif "__launch_bounds__" in l:
            m2 = re.search(r"void\s+(\w+)\s*\(", l)
            if m2:
                out.append((i, m2.group(1)))
    return out, lines


def classify(op):
    if op.startswith("v_"):
        if re.match(r"v_(rcp|rsq|sqrt|exp|log|sin|cos)_", op):
            return "valu_trans"
        if "_f64" in op or op.startswith("v_div_") or op.startswith("v_frexp") or op.startswith("v_ldexp"):
            return "valu_f64"
        if re.match(r"v_(mul_lo|mul_hi|mad_u64|mad_i64)", op):
            return "valu_imul"
        return "valu_other"
    if op.startswith("s_"):
        return "wait" if op.startswith("s_waitcnt") or op.startswith("s_nop") else "salu"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("ds_"):
        return "lds"
    return "other"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kernel", default="k_waveILb0ELi0E", help="substring of the mangled kernel name")
    ap.add_argument("--trips", default="", help="loop trip counts per wave, e.g. inner=12.3,leaf=3.1,reject=5.0")
    ap.add_argument("--asm", default=None, help="an existing .s made with -gline-tables-only (default: compile now)")
    ap.add_argument("--by-line", action="store_true", help="also list the 40 source lines with most vector instructions")
    a = ap.parse_args()

    asm = a.asm
    if not asm:
        asm = os.path.join(tempfile.gettempdir(), "rt_device_census.s")
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math",
               "-gline-tables-only", "--cuda-device-only", "-S", "-o", asm, "rt_device.hip"]
        subprocess.run(cmd, cwd=CSRC, check=True, stderr=subprocess.DEVNULL)

    src = {}
    for name in ("rt_kernels.hip.h", "rt_device.hip"):
        src[name] = function_starts(os.path.join(CSRC, name))
    # part markers -> line numbers
    part_lines = collections.defaultdict(list)  # (file, function) -> [(line, part)]
    for fname, func, marker, part in PARTS:
        starts, lines = src[fname]
        f_lines = [l for l, n in starts if n == func]
        if not f_lines:
            continue
        f0 = f_lines[0]
        nxt = [l for l, n in starts if l > f0]
        f1 = nxt[0] if nxt else len(lines) + 1
        for i in range(f0, f1):
            if re.search(marker, lines[i - 1]):
                part_lines[(fname, func)].append((i, part))
                break
    for k in part_lines:
        part_lines[k].sort()

    files = {}
    counts = collections.defaultdict(collections.Counter)
    by_line = collections.Counter()
    in_kernel = False
    cur = ("?", 0)
    for l in open(asm):
        m = re.match(r"\s*\.file\s+(\d+)\s+\"[^\"]*\"\s+\"([^\"]+)\"", l) or re.match(r"\s*\.file\s+(\d+)\s+\"([^\"]+)\"", l)
        if m:
            files[int(m.group(1))] = os.path.basename(m.group(2))
            continue
        if re.match(r"^_Z\w+:", l):
            in_kernel = a.kernel in l
            continue
        if not in_kernel:
            continue
        if ".end_amdhsa_kernel" in l or l.strip().startswith(".section"):
            in_kernel = False
            continue
        m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", l)
        if m:
            cur = (files.get(int(m.group(1)), "?"), int(m.group(2)))
            continue
        t = l.strip()
        if not t or t.startswith((".", ";")) or t.endswith(":"):
            continue
        op = t.split()[0]
        cls = classify(op)
        fname, line = cur
        where = fname
        if fname in src:
            starts, _ = src[fname]
            i = bisect.bisect_right([s for s, _ in starts], line) - 1
            func = starts[i][1] if i >= 0 else "?"
            where = func
            pl = part_lines.get((fname, func))
            if pl:
                j = bisect.bisect_right([s for s, _ in pl], line) - 1
                where = pl[j][1] if j >= 0 else func + ": prologue"
        counts[where][cls] += 1
        if cls.startswith("valu"):
            by_line[(fname, line)] += 1

    trips = {}
    for kv in filter(None, a.trips.split(",")):
        k, v = kv.split("=")
        trips[k] = float(v)
    cols = ["valu_f64", "valu_trans", "valu_imul", "valu_other", "salu", "vmem", "lds", "wait"]
    print(f"kernel *{a.kernel}*: static instruction counts by source part")
    print(f"{'part':52s}" + "".join(f"{c:>11s}" for c in cols) + f"{'VALU':>8s}")
    tot = collections.Counter()
    rows = sorted(counts.items(), key=lambda kv: -sum(v for k, v in kv[1].items() if k.startswith("valu")))
    for where, c in rows:
        valu = sum(v for k, v in c.items() if k.startswith("valu"))
        print(f"{where[:52]:52s}" + "".join(f"{c.get(k, 0):11d}" for k in cols) + f"{valu:8d}")
        tot.update(c)
    valu = sum(v for k, v in tot.items() if k.startswith("valu"))
    print(f"{'TOTAL':52s}" + "".join(f"{tot.get(k, 0):11d}" for k in cols) + f"{valu:8d}")
    if a.by_line:
        print("\nsource lines with most vector instructions:")
        for (fname, line), n in by_line.most_common(40):
            text = ""
            if fname in src:
                text = src[fname][1][line - 1].strip()[:100]
            print(f"  {n:4d}  {fname}:{line}  {text}")


if __name__ == "__main__":
    main()
