#!/bin/bash
# 2 ranks sharing the one-GPU box's card, gloo carrying the all-reduce: the ORDER of bench.py --gpus N (baseline written out first,
# guarded experiments, watchdog), not a number. Usage: tools/rehearse_dp2.sh <outdir>
set -o pipefail
out=${1:-gpurun_out/r4/dp2}; mkdir -p $out
export MST_FORCE_DEVICE=0 MST_DIST_BACKEND=gloo
MST_BENCH_PARTIAL=$out/partial_ok.json timeout -k 10 300 python bench.py --gpus 2 --steps 20 --warmup 5 > $out/ok.json 2> $out/ok.err || exit 1
MST_BENCH_PARTIAL=$out/partial_raise.json MST_BENCH_TEST_CANDIDATE=raise timeout -k 10 300 python bench.py --gpus 2 --steps 20 --warmup 5 > $out/raise.json 2> $out/raise.err || exit 2
MST_BENCH_PARTIAL=$out/partial_hang.json MST_BENCH_TEST_CANDIDATE=hang MST_BENCH_EXPERIMENT_TIMEOUT=20 timeout -k 10 300 python bench.py --gpus 2 --steps 20 --warmup 5 > $out/hang.json 2> $out/hang.err || exit 3
MST_BENCH_PARTIAL=$out/partial_tune.json MST_RCCL_AUTOTUNE=1 timeout -k 10 300 python bench.py --gpus 2 --steps 20 --warmup 5 > $out/tune.json 2> $out/tune.err || exit 4
for f in ok raise hang tune; do python - $out/$f.json <<'PY'
import json,sys
j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r=j["rccl"]
print(sys.argv[1], "value", round(j["value"]), "ms", round(j["ms_per_step"],3), "| schedule:", r.get("schedule"), "| experiments:", json.dumps(r.get("experiments"))[:600])
PY
done
