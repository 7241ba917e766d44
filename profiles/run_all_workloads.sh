#!/bin/bash
# The five profiling passes of profiles/run_profile.sh for a list of bench.py workloads, one after
# the other. Usage (through gpurun, from the repo root):
#   bash profiles/run_all_workloads.sh <tag> <workload> [<workload> ...]
# Output: gpurun_out/prof_<tag>_<workload>/; reduce each with tools/pmc_summary.py, then
# tools/make_hbm_traffic.py.
set -e
TAG=$1; shift
for w in "$@"; do
  case $w in
    sphere_floor_1080p_64spp|spot_default_10spp|spot_20k_default_10spp) extra="";;
    *) extra="--steps 3 --warmup 1";;   # frames of 0.1 .. 1.4 s: three timed frames are enough for per-launch means
  esac
  echo "== $w"
  BENCH_ARGS="--workload $w $extra" bash profiles/run_profile.sh ${TAG}_$w > /dev/null
  tail -c 300 gpurun_out/prof_${TAG}_$w/trace.log | tr '\n' ' ' | cut -c1-200
  echo
done
