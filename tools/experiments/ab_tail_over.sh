for v in 8 10 12 16; do
  export MST_EXTRA_FLAGS="row_tail.hip=-DMST_TAIL_OVERSUBSCRIBE=$v"
  python -m musicstyletransfer_amd.csrc.build --force > /dev/null 2>&1 || { echo build failed; exit 1; }
  echo "== oversubscribe $v: $(python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k row_tail 2>&1 | tail -1)"
  python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['ms_per_step_median'])"
done
unset MST_EXTRA_FLAGS
python -m musicstyletransfer_amd.csrc.build --force > /dev/null 2>&1
