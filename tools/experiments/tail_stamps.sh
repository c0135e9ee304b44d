export MST_EXTRA_FLAGS="row_tail.hip=-DMST_TAIL_STAMPS"
python -m musicstyletransfer_amd.csrc.build --force > /dev/null 2>&1 || { echo build failed; exit 1; }
python tools/bench_tail_stamps.py | tail -3
python tools/bench_tail_bwd_stamps.py | tail -4
unset MST_EXTRA_FLAGS
python -m musicstyletransfer_amd.csrc.build --force > /dev/null 2>&1
