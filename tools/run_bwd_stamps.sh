#!/bin/bash
# on the GPU box: stamps build of attention.hip, the three backward shapes, then the product build again
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
touch musicstyletransfer_amd/csrc/attention.hip
MST_EXTRA_FLAGS="attention.hip=-DMST_ATT_STAMPS" python -m musicstyletransfer_amd.csrc.build > /dev/null
for m in enc sparse dec; do python tools/bench_attn_bwd_stamps.py $m; done 2>&1 | tee gpurun_out/attn_bwd_stamps.txt
touch musicstyletransfer_amd/csrc/attention.hip
python -m musicstyletransfer_amd.csrc.build > /dev/null
python tools/bench_attn.py 30 2>&1 | tee gpurun_out/bench_attn.txt
