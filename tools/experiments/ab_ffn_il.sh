python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "ffn" 2>&1 | tail -2
for v in "-DMST_FFN_IL=1" "-DMST_FFN_IL=0" "-DMST_FFN_IL=1"; do
  export MST_EXTRA_FLAGS="gemm_nt.hip=$v"
  python -m musicstyletransfer_amd.csrc.build --force > /dev/null 2>&1 || { echo build failed; exit 1; }
  echo "== $v"; python tools/bench_ffn.py
  python bench.py --steps 100 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['ms_per_step'], d['ms_per_step_median'])
"
done
export MST_EXTRA_FLAGS="gemm_nt.hip=-DMST_FFN_STAMPS"
python -m musicstyletransfer_amd.csrc.build --force > /dev/null 2>&1 || { echo build failed; exit 1; }
python tools/bench_ffn_stamps.py | tail -13
unset MST_EXTRA_FLAGS
python -m musicstyletransfer_amd.csrc.build --force > /dev/null 2>&1
