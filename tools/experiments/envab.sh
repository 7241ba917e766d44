#!/bin/bash
# A/B of bench.py workloads under different environment settings or library variants, on the GPU box:
#   bash tools/experiments/envab.sh <out dir under gpurun_out/> <workloads...> -- "<label>:<ENV=V ENV2=V2>" ...
# ("base:" = defaults; RT_AMD_LIB=$PWD/exp_build/librt_<name>.so selects a variant built by build_variant.sh).
# Prints one line per run: ms per frame, Mrays/s, and the kernels' ms per frame.
OUT=$1; shift
WL=(); while [ "$1" != "--" ]; do WL+=("$1"); shift; done; shift
mkdir -p gpurun_out/$OUT
for spec in "$@"; do
  label=${spec%%:*}; envs=${spec#*:}
  for w in "${WL[@]}"; do
    steps=5; case $w in sphere_floor*|spot_default*|spot_20k*) steps=40;; esac
    env $envs timeout -k 10 200 python bench.py --workload $w --steps $steps --warmup 3 --no-cpu-baseline --no-secondary > gpurun_out/$OUT/${label}_$w.json 2> gpurun_out/$OUT/${label}_$w.err || { echo "FAILED $label $w"; tail -5 gpurun_out/$OUT/${label}_$w.err; exit 1; }
    python - gpurun_out/$OUT/${label}_$w.json $label $w <<'PY'
import json, sys
j = json.load(open(sys.argv[1]))
k = "  ".join(f"{k['kernel'].split(' ')[0].replace('k_wave<false,','w<')} x{k['launches_per_frame']:g} {k['ms_per_frame']:.2f}" for k in j['kernels'])
print(f"{sys.argv[2]:12s} {sys.argv[3]:26s} {j['ms_per_step']:9.3f} ms {j['value']:9.1f}  {k}")
PY
  done
done
