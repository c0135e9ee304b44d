for v in "" "attention.hip=-DMST_ATT16_WAVES_BWD=4" "attention.hip=-DMST_ATT16_WAVES_BWD=4 -DMST_ATT16_WAVES_FWD=4"; do
  export MST_EXTRA_FLAGS="$v"
  python -m musicstyletransfer_amd.csrc.build --force > /dev/null 2>&1 || { echo build failed; exit 1; }
  echo "== flags: $v"
  python tools/bench_attn.py 50 | tail -1
  python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['ms_per_step_median'])"
done
unset MST_EXTRA_FLAGS
python -m musicstyletransfer_amd.csrc.build --force > /dev/null 2>&1
python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k attn 2>&1 | tail -2
