python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "wgrad" 2>&1 | tail -1
for v in 1 0 1 0; do
  MST_WGRAD_MIXED=$v python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('mixed $v', d['ms_per_step'], d['ms_per_step_median'], round(d['roofline']['families'][0]['avg_launch_ms']*1e3,1))"
done
python -m pytest tests/test_step_gpu.py tests/test_configs_gpu.py -x -q -m gpu 2>&1 | tail -1
