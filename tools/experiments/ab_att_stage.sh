#!/bin/bash
# streaming attention kernels: rows staged per step (MST_ATT_STAGE), configs[4] (T 1024): per-kernel times under rocprofv3
cd $GRAFT_REPO_ROOT
for st in 64 128 256; do
  touch musicstyletransfer_amd/csrc/attention.hip
  MST_EXTRA_FLAGS="attention.hip=-DMST_ATT_STAGE=$st" python -m musicstyletransfer_amd.csrc.build > /dev/null 2>&1 || { echo "build failed ($st)"; exit 1; }
  bash tools/profile_config.sh 4 > /dev/null 2>&1
  echo "== stage $st"; grep "attn_\|total" gpurun_out/cfg4/timeline.txt
  rm -rf gpurun_out/cfg4/stats
done
touch musicstyletransfer_amd/csrc/attention.hip
python -m musicstyletransfer_amd.csrc.build > /dev/null 2>&1
