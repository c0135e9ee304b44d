#!/bin/bash
# streaming attention kernels: rows staged per step (MST_ATT_STAGE), configs[4] (T 1024)
cd $GRAFT_REPO_ROOT
for st in 64 128 256; do
  touch musicstyletransfer_amd/csrc/attention.hip
  MST_EXTRA_FLAGS="attention.hip=-DMST_ATT_STAGE=$st" python -m musicstyletransfer_amd.csrc.build > /dev/null 2>&1 || { echo "build failed ($st)"; exit 1; }
  echo -n "stage $st: "
  python bench.py --config 4 --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernel'][:60], d['roofline']['avg_launch_ms'])"
done
touch musicstyletransfer_amd/csrc/attention.hip
python -m musicstyletransfer_amd.csrc.build > /dev/null 2>&1
