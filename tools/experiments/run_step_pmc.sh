#!/bin/bash
# usage (GPU box): tools/experiments/run_step_pmc.sh [bench args]  -> per-kernel SQ counters of the captured step (medians over launches):
# VALU / LDS / MFMA activity as fractions of the launch's SIMD cycles (time x ~1.9 GHz x 1024 SIMDs is not known per launch, so
# fractions are given against SQ_BUSY_CYCLES-derived wave cycles: active quad-cycles x 4 / (duration_us x 1900 x 1024))
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/step_pmc
rm -rf $out; mkdir -p $out
B="$GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline $@"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVES --output-format csv -d $out/a -o r -- python3 $B > /dev/null 2> $out/a.err
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $out/b -o r -- python3 $B > /dev/null 2> $out/b.err
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, collections, glob, os
out = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/step_pmc")
acc = collections.defaultdict(lambda: collections.defaultdict(list))
order = []
for sub in "ab":
    f = glob.glob(f"{out}/{sub}/*counter_collection.csv")
    if not f:
        print(sub, "no counter file", open(f"{out}/{sub}.err").read()[-400:]); continue
    for r in csv.DictReader(open(f[0])):
        k = (r["Kernel_Name"][:64], r["Grid_Size"])
        if k not in order: order.append(k)
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
t = glob.glob(f"{out}/a/*kernel_trace.csv")
dur = collections.defaultdict(list)
for r in csv.DictReader(open(t[0])):
    dur[r["Kernel_Name"][:64]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
med = lambda v: sorted(v)[len(v) // 2]
lines = ["kernel grid us | per wave: quad-cycles, VALU insts (excl. MFMA), MFMA insts, LDS insts | of wave cycles: wait_any, wait_inst, active_valu, active_lds | MFMA-busy share of SIMD time"]
for k in order:
    c = {n: med(v) for n, v in acc[k].items()}
    if "SQ_WAVES" not in c or c["SQ_WAVES"] == 0 or "SQ_INSTS_VALU" not in c: continue
    w, wc = c["SQ_WAVES"], c["SQ_WAVE_CYCLES"]
    us = med(dur[k[0]]) if k[0] in dur else 0
    simd_cycles = us * 1900 * 1024
    lines.append(f"{k[0][:58]:58s} {k[1]:>8s} {us:6.1f} | {wc / w:8.0f} {(c['SQ_INSTS_VALU'] - c['SQ_INSTS_MFMA']) / w:7.0f} {c['SQ_INSTS_MFMA'] / w:6.0f} {c['SQ_INSTS_LDS'] / w:6.0f} | "
                 f"{c['SQ_WAIT_ANY'] / wc:5.2f} {c['SQ_WAIT_INST_ANY'] / wc:5.2f} {c['SQ_ACTIVE_INST_VALU'] / wc:5.2f} {c['SQ_ACTIVE_INST_LDS'] / wc:5.2f} | "
                 f"{c['SQ_VALU_MFMA_BUSY_CYCLES'] / simd_cycles if simd_cycles else 0:5.2f} valu-active {c['SQ_ACTIVE_INST_VALU'] * 4 / simd_cycles if simd_cycles else 0:5.2f}")
open(f"{out}/summary.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
rm -rf $out/a $out/b
