#!/bin/bash
# The five profiling passes of profiles/run_profile.sh for a list of bench.py workloads, one after
# the other. Usage (through gpurun, from the repo root):
#   bash profiles/run_all_workloads.sh <tag> <workload> [<workload> ...]
# Output: gpurun_out/sum_<tag>/<workload>_{kernel_stats.csv,pmc_summary.json} (tools/pmc_summary.py,
# run here) and the logs under gpurun_out/prof_<tag>_<workload>/; afterwards copy the summaries into
# profiles/ and run tools/make_hbm_traffic.py on each.
set -e
TAG=$1; shift
for w in "$@"; do
  case $w in
    sphere_floor_1080p_64spp|spot_default_10spp|spot_20k_default_10spp) extra="";;
    *) extra="--steps 3 --warmup 1";;   # frames of 0.1 .. 1.4 s: three timed frames are enough for per-launch means
  esac
  echo "== $w"
  BENCH_ARGS="--workload $w $extra" bash profiles/run_profile.sh ${TAG}_$w > /dev/null
  # reduce here and drop the raw counter files: what a call may bring back is bounded
  mkdir -p gpurun_out/sum_$TAG
  python tools/pmc_summary.py gpurun_out/prof_${TAG}_$w gpurun_out/sum_$TAG/$w > /dev/null
  find gpurun_out/prof_${TAG}_$w -name "*.csv" -delete
  ls gpurun_out/sum_$TAG | grep -c $w
done
