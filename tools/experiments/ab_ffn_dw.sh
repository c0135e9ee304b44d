export MST_EXTRA_FLAGS="gemm_nt.hip=-DMST_FFN_DW_RING=4 -DMST_FFN_DW_WGN256=8 -DMST_FFN_DW_FAKE"
python -m musicstyletransfer_amd.csrc.build --force > /dev/null 2>&1 || { echo build failed; exit 1; }
echo "== DW fake ring 4 waves 8"; python tools/bench_ffn_f.py; python tools/bench_ffn_m.py
echo "== staged"; MST_FFN_STAGED=1 python tools/bench_ffn_f.py
unset MST_EXTRA_FLAGS
python -m musicstyletransfer_amd.csrc.build --force > /dev/null 2>&1
