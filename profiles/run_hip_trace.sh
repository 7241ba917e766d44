#!/bin/bash
# HIP API trace of a few frames of bench.py (no counters): shows what the host calls between the
# first-stage launch of a frame and its resolve. Usage: bash profiles/run_hip_trace.sh <tag>
set -e
TAG=${1:-hiptrace}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="$R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-secondary ${BENCH_ARGS:-}"
timeout -s KILL 240 rocprofv3 --hip-trace --kernel-trace --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/trace.log 2>&1
find $OUT -name "*.csv" | head
