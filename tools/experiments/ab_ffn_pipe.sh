set -e
mkdir -p gpurun_out/r4
export MST_EXTRA_FLAGS="gemm_nt.hip=-DMST_FFN_PIPE=1"
python -m musicstyletransfer_amd.csrc.build --force > gpurun_out/r4/pipe_build.log 2>&1
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -q -x -k "ffn_ln or proj_ffn" > gpurun_out/r4/pipe_tests.log 2>&1 || { tail -30 gpurun_out/r4/pipe_tests.log; exit 1; }
tail -2 gpurun_out/r4/pipe_tests.log
echo "PIPE on:"; timeout -k 10 200 python tools/bench_ffn.py
unset MST_EXTRA_FLAGS
python -m musicstyletransfer_amd.csrc.build --force >> gpurun_out/r4/pipe_build.log 2>&1
echo "PIPE off:"; timeout -k 10 200 python tools/bench_ffn.py
