#!/bin/bash
# Runs bench.py on every workload and keeps the JSON lines. Usage (through gpurun, from the repo
# root): bash profiles/run_configs.sh <out dir under gpurun_out/> [workload ...]
set -e
OUT=gpurun_out/${1:-configs}
shift || true
mkdir -p $OUT
W=${@:-"sphere_floor_1080p_64spp spot_default_10spp spot_20k_default_10spp spot_4k_256spp ogre_4k_256spp spot_grid14_8k_4spp spot_4k_1024spp"}
for w in $W; do
  steps=5; [ $w = sphere_floor_1080p_64spp ] && steps=40; [ $w = spot_default_10spp ] && steps=40; [ $w = spot_20k_default_10spp ] && steps=40; [ $w = spot_4k_1024spp ] && steps=2
  timeout -k 10 400 python bench.py --workload $w --steps $steps --warmup 2 --no-cpu-baseline --no-secondary > $OUT/$w.json 2> $OUT/$w.err
  python - $OUT/$w.json <<'PY'
import json, sys
j = json.load(open(sys.argv[1]))
print(f"{j['config']['workload'].split(' ')[0]:28s} {j['ms_per_step']:9.3f} ms  {j['value']:9.1f} Mrays/s  " + "  ".join(f"{k['kernel'].split(' ')[0]} x{k['launches_per_frame']:g} {k['ms_per_frame']:.2f}" for k in j['kernels']))
PY
done
