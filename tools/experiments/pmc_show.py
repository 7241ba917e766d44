"""Reduces the output of tools/experiments/pmc_sq.sh: one line per k_wave kernel."""
import csv, glob, os, sys
src = sys.argv[1]
dur = {}
for f in glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        name = row["Name"].split("(")[0].replace("void ", "").strip()
        dur[name] = (int(row["Calls"]), float(row["TotalDurationNs"]) / 1e6)
out = {}
for f in glob.glob(os.path.join(src, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"].split("(")[0].replace("void ", "").strip()
        c = out.setdefault(name, {})
        c[row["Counter_Name"]] = c.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
for name in sorted(out):
    if not (name.startswith("k_wave") or name.startswith("k_pool")): continue
    c = out[name]; calls, ms = dur.get(name, (0, 0.0))
    w = max(1.0, c.get("SQ_WAVES", 0))
    cyc = ms * 1e-3 * 2.4e9 * 1024  # SIMD-cycles at 2.4 GHz
    print(f"{name:40s} calls {calls:4d} {ms:9.2f} ms | waves {w:.3g} valu/wave {c.get('SQ_INSTS_VALU',0)/w:7.0f} salu/wave {c.get('SQ_INSTS_SALU',0)/w:6.0f} vmem_rd/wave {c.get('SQ_INSTS_VMEM_RD',0)/w:5.0f} vmem_wr/wave {c.get('SQ_INSTS_VMEM_WR',0)/w:5.0f} lds/wave {c.get('SQ_INSTS_LDS',0)/w:5.0f}"
          f" | valu busy {4*c.get('SQ_ACTIVE_INST_VALU',0)/max(cyc,1):.2f} (4*ACTIVE_INST_VALU/SIMD-cycles) lanes {c.get('SQ_THREAD_CYCLES_VALU',0)/max(1,64*c.get('SQ_ACTIVE_INST_VALU',0)):.2f}"
          f" | wave-cycles: active_any {c.get('SQ_ACTIVE_INST_ANY',0)/max(1,c.get('SQ_WAVE_CYCLES',0)):.2f} wait_inst {c.get('SQ_WAIT_INST_ANY',0)/max(1,c.get('SQ_WAVE_CYCLES',0)):.2f} wait_any {c.get('SQ_WAIT_ANY',0)/max(1,c.get('SQ_WAVE_CYCLES',0)):.2f} occupancy {4*c.get('SQ_WAVE_CYCLES',0)/max(cyc,1):.2f}")
