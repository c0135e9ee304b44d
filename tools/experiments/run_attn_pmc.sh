#!/bin/bash
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/attn_pmc
rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $out/a -o r -- python3 $GRAFT_REPO_ROOT/tools/experiments/attn_pmc.py > /dev/null 2> $out/a.err
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_WAVES --output-format csv -d $out/b -o r -- python3 $GRAFT_REPO_ROOT/tools/experiments/attn_pmc.py > /dev/null 2> $out/b.err
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, collections, glob, os
out = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/attn_pmc")
for sub in ("a", "b"):
    f = glob.glob(f"{out}/{sub}/*counter_collection.csv")
    if not f:
        print(sub, "no counter file", open(f"{out}/{sub}.err").read()[-600:]); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"]
        if "attn_" not in k: continue
        acc[(k[:60], r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for key, cs in acc.items():
        print(key[0], key[1], {c: round(sorted(v)[len(v) // 2]) for c, v in cs.items()})
PY
rm -rf $out/a $out/b
