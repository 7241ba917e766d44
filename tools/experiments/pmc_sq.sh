#!/bin/bash
# Kernel trace and two passes of SQ counters of one bench.py workload, reduced to a line per kernel
# (instructions per wave, issue-port share, lane use, where the waves' cycles go):
#   bash tools/experiments/pmc_sq.sh <tag> <workload> [steps]   (environment passes through) -> gpurun_out/pmcx_<tag>/summary.txt
TAG=$1; WL=$2; STEPS=${3:-2}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmcx_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="$R/bench.py --no-cpu-baseline --no-secondary --workload $WL --steps $STEPS --warmup 2"
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/trace.log 2>&1 || { echo trace failed; tail -5 $OUT/trace.log; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/pmc_sq -- python3 $ARGS > $OUT/pmc_sq.log 2>&1 || { echo pmc failed; tail -5 $OUT/pmc_sq.log; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/pmc_sq2 -- python3 $ARGS > $OUT/pmc_sq2.log 2>&1 || { echo pmc2 failed; tail -5 $OUT/pmc_sq2.log; exit 1; }
cd $R && python3 tools/experiments/pmc_show.py $OUT > $OUT/summary.txt && cat $OUT/summary.txt
find $OUT -name "*.csv" -size +2M -delete
