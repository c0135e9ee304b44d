for v in "-DMST_WGRAD_IL=0" "-DMST_WGRAD_IL=1"; do
  export MST_EXTRA_FLAGS="gemm_wgrad.hip=-DMST_WGRAD_STAMPS $v"
  python -m musicstyletransfer_amd.csrc.build --force > /dev/null 2>&1 || { echo build failed; exit 1; }
  echo "== $v"; python tools/bench_wgrad_stamps.py | tail -4
done
unset MST_EXTRA_FLAGS
python -m musicstyletransfer_amd.csrc.build --force > /dev/null 2>&1
python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "wgrad" 2>&1 | tail -2
python bench.py --steps 100 --warmup 20 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/b.json; python -c "
import json
d=json.loads(open('gpurun_out/b.json').read())
print(d['ms_per_step'], d['ms_per_step_median'])
for f in d['roofline']['families']: print(f['kernel'][:60], round(f['avg_launch_ms']*1e3,1))
"
