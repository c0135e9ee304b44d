#!/bin/bash
# usage (GPU box): tools/experiments/run_ffn_pmc.sh  -> per-kernel medians of SQ / TCP / TCC counters for the ffn_ln launches
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/ffn_pmc
rm -rf $out; mkdir -p $out
P=$GRAFT_REPO_ROOT/tools/experiments/ffn_pmc.py
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $out/a -o r -- python3 $P > /dev/null 2> $out/a.err
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_WAVES --output-format csv -d $out/b -o r -- python3 $P > /dev/null 2> $out/b.err
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL --output-format csv -d $out/c -o r -- python3 $P > /dev/null 2> $out/c.err
rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum --output-format csv -d $out/d -o r -- python3 $P > /dev/null 2> $out/d.err
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum --output-format csv -d $out/e -o r -- python3 $P > /dev/null 2> $out/e.err
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d $out/f -o r -- python3 $P > /dev/null 2> $out/f.err
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, collections, glob, os
out = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/ffn_pmc")
for sub in "abcdef":
    f = glob.glob(f"{out}/{sub}/*counter_collection.csv")
    if not f:
        print(sub, "no counter file", open(f"{out}/{sub}.err").read()[-400:]); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"]
        if "ffn_ln" not in k: continue
        acc[(k[:70], r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for key, cs in acc.items():
        print(sub, key[0][-40:], key[1], {c: round(sorted(v)[len(v) // 2]) for c, v in cs.items()})
    t = glob.glob(f"{out}/{sub}/*kernel_trace.csv")
    if t and sub == "a":
        d = collections.defaultdict(list)
        for r in csv.DictReader(open(t[0])):
            if "ffn_ln" in r["Kernel_Name"]:
                d[r["Kernel_Name"][:70][-40:]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        for k, v in d.items(): print("time", k, f"{sorted(v)[len(v)//2]:.1f} us")
PY
rm -rf $out/a $out/b $out/c $out/d $out/e $out/f
