python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm" 2>&1 | tail -1
for v in 1 0 1 0; do
  MST_GEMM_BK32=$v python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('bk32 $v', d['ms_per_step'], d['ms_per_step_median'])"
done
