run() { python bench.py --config 4 --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], [ (f['kernel'][:22], round(f['avg_launch_ms']*1e3,1)) for f in d['roofline']['families']])"; }
run default
MST_FUSE_PROJ=0 run noproj
export MST_EXTRA_FLAGS="gemm_wgrad.hip=-DMST_WGRAD_IL=0"
python -m musicstyletransfer_amd.csrc.build --force > /dev/null 2>&1
run wgrad_il0
export MST_EXTRA_FLAGS="gemm_nt.hip=-DMST_FFN_IL=0"
python -m musicstyletransfer_amd.csrc.build --force > /dev/null 2>&1
run ffn_il0
unset MST_EXTRA_FLAGS
python -m musicstyletransfer_amd.csrc.build --force > /dev/null 2>&1
