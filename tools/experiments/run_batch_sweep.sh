#!/bin/bash
cd /tmp && export TMPDIR=/tmp
for B in 64 32 16; do
  rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/bsweep/b$B -o r -- python3 $GRAFT_REPO_ROOT/tools/experiments/batch_sweep.py $B 60 2>&1 | grep "ms per step"
done
cd $GRAFT_REPO_ROOT
python3 tools/step_timeline.py gpurun_out/bsweep/b64/r_kernel_trace.csv gpurun_out/bsweep/b32/r_kernel_trace.csv | tee gpurun_out/bsweep/t64_32.txt
python3 tools/step_timeline.py gpurun_out/bsweep/b32/r_kernel_trace.csv gpurun_out/bsweep/b16/r_kernel_trace.csv | tee gpurun_out/bsweep/t32_16.txt
rm -f gpurun_out/bsweep/*/r_kernel_trace.csv gpurun_out/bsweep/*/*.db
