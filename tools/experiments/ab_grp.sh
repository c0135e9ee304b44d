python -m pytest tests/test_kernels_gpu.py tests/test_step_gpu.py -x -q -m gpu -k "gemm or step" 2>&1 | tail -1
for i in 1 2; do python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['ms_per_step_median'])"; done
tools/experiments/trace_step.sh grp1; sed -n 1,3p gpurun_out/grp1_timeline.txt; sed -n 10,11p gpurun_out/grp1_timeline.txt
