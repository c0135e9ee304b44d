export MST_EXTRA_FLAGS="gemm_nt.hip=-DMST_FFN_STAMPS"
python -m musicstyletransfer_amd.csrc.build --force > /dev/null 2>&1 || { echo build failed; exit 1; }
python tools/bench_ffn_stamps.py | tail -7 | head -3
python tools/bench_ffn_bwd_stamps.py | tail -8
unset MST_EXTRA_FLAGS
python -m musicstyletransfer_amd.csrc.build --force > /dev/null 2>&1
