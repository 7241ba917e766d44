#!/bin/bash
# The round's record runs, one after the other (through gpurun, from the repo root):
#   bash profiles/run_round.sh <tag>     -> gpurun_out/<tag>/
# GPU tests, bench.py on every workload (pipelined), the default bench line with its CPU baseline,
# and the frames of configs[0] one after the other (each waited for, as a display loop would).
TAG=${1:-round}
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $OUT/gpu_tests.log 2>&1 || { tail -20 $OUT/gpu_tests.log; exit 1; }
tail -2 $OUT/gpu_tests.log
bash profiles/run_configs.sh $TAG/configs || exit 1
timeout -k 10 400 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err || { tail -5 $OUT/bench_default.err; exit 1; }
python - $OUT/bench_default.json <<'PY'
import json, sys
j = json.load(open(sys.argv[1]))
print("default line:", j["ms_per_step"], "ms", j["value"], j["unit"], "roofline", j["roofline"]["bound"], j["roofline"]["frac"], j["roofline"]["counters"],
      "| cpu", j["cpu_baseline"]["value"], "| secondary", j["secondary"]["value"], "s")
PY
RT_BENCH_NO_PIPELINE=1 timeout -k 10 300 python bench.py --workload spot_default_10spp --steps 40 --no-cpu-baseline --no-secondary > $OUT/spot_default_one_after_the_other.json 2> /dev/null || exit 1
python - $OUT/spot_default_one_after_the_other.json <<'PY'
import json, sys
j = json.load(open(sys.argv[1]))
print("configs[0], frames one after the other:", j["ms_per_step"], "ms;", j["config"]["frames"])
PY
