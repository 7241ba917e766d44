#!/bin/bash
# Profiles bench.py's default workload on the GPU box. Usage (through gpurun, from the repo root):
#   bash profiles/run_profile.sh <tag>          (BENCH_ARGS="--workload ..." adds bench.py options)
# then, anywhere:  python tools/pmc_summary.py gpurun_out/prof_<tag> profiles/<round>/<tag>
# Writes rocprofv3 output under gpurun_out/prof_<tag>/; copy the summaries into profiles/.
set -e
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="$R/bench.py --no-cpu-baseline --no-secondary ${BENCH_ARGS:-}"  # bench.py's own defaults for steps and warm-up; the CPU leg and the
# Spot 4K frames it times after the main loop are left out (they would mix another workload into the per-kernel means)
timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/trace.log 2>&1
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ARGS > $OUT/pmc_fetch.log 2>&1
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ARGS > $OUT/pmc_write.log 2>&1
timeout -s KILL 240 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/pmc_sq -- python3 $ARGS > $OUT/pmc_sq.log 2>&1
timeout -s KILL 240 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_sq2 -- python3 $ARGS > $OUT/pmc_sq2.log 2>&1
timeout -s KILL 240 rocprofv3 --pmc TA_BUSY_avr TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_tcp -- python3 $ARGS > $OUT/pmc_tcp.log 2>&1
find $OUT -name "*.csv" | head -50
