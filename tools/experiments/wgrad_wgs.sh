export MST_EXTRA_FLAGS="gemm_wgrad.hip=-DMST_WGRAD_STAMPS"
python -m musicstyletransfer_amd.csrc.build --force > /dev/null 2>&1 || { echo build failed; exit 1; }
MST_CLS_WGRAD=1 python tools/bench_wgrad_wgs.py
echo "== without the class problem"; MST_CLS_WGRAD=0 python tools/bench_wgrad_wgs.py
unset MST_EXTRA_FLAGS
python -m musicstyletransfer_amd.csrc.build --force > /dev/null 2>&1
