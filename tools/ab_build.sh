#!/bin/bash
# usage (on the GPU box): tools/ab_build.sh "<file.hip>=-DMST_EXP_A" [rounds]   -> alternates builds with / without the flag and benches each
# (numbers from different gpurun calls differ by +-1.5 %; an A/B has to run inside one call)
flag="$1"; rounds="${2:-2}"
for r in $(seq $rounds); do
  for v in A B; do
    if [ $v = A ]; then export MST_EXTRA_FLAGS="$flag"; else unset MST_EXTRA_FLAGS; fi
    python -m musicstyletransfer_amd.csrc.build --force > /dev/null 2>&1 || { echo "build failed ($v)"; exit 1; }
    echo -n "$v (flag $( [ $v = A ] && echo on || echo off )): "
    python bench.py --steps 200 --warmup 20 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
  done
done
unset MST_EXTRA_FLAGS
python -m musicstyletransfer_amd.csrc.build --force > /dev/null 2>&1
