for v in 0 1 2; do
  export MST_EXTRA_FLAGS="gemm_wgrad.hip=-DMST_WGRAD_NARROW_SPLIT=$v"
  python -m musicstyletransfer_amd.csrc.build --force > /dev/null 2>&1 || { echo build failed; exit 1; }
  tools/experiments/trace_step.sh sp$v
  echo "== variant $v: $(grep -h 'wgrad_kernel\|wgrad_reduce\|total' gpurun_out/sp${v}_timeline.txt | tr '\n' ' ')"
done
unset MST_EXTRA_FLAGS
python -m musicstyletransfer_amd.csrc.build --force > /dev/null 2>&1
