python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "attention or ffn" 2>&1 | tail -1
for v in 1 0 1 0; do
  export MST_EXTRA_FLAGS="gemm_nt.hip=-DMST_XCD_ROWS=$v;attention.hip=-DMST_XCD_ROWS=$v"
  python -m musicstyletransfer_amd.csrc.build --force > /dev/null 2>&1 || { echo build failed; exit 1; }
  python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('xcd rows $v', d['ms_per_step'], d['ms_per_step_median'])"
done
unset MST_EXTRA_FLAGS
python -m musicstyletransfer_amd.csrc.build --force > /dev/null 2>&1
