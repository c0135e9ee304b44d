for v in 1 0 1 0; do
  export MST_EXTRA_FLAGS="gemm_nt.hip=-DMST_FFN_COPY_IL=$v"
  python -m musicstyletransfer_amd.csrc.build --force > /dev/null 2>&1 || { echo build failed; exit 1; }
  echo "== copy_il $v: $(python tools/bench_ffn.py | grep forward | tr '\n' ' ')"
  python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['ms_per_step_median'])"
done
unset MST_EXTRA_FLAGS
python -m musicstyletransfer_amd.csrc.build --force > /dev/null 2>&1
