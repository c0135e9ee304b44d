#!/bin/bash
# In-call A/B of prebuilt library variants (tools/experiments/lib<X>.so): each is copied over the product library and
# bench.py is run for configs[1] and configs[4]; two interleaved rounds. Usage: ab_libs.sh H A B C
set -u
OUT=gpurun_out/ab_libs.txt
: > $OUT
for rep in 1 2; do
  for v in "$@"; do
    cp tools/experiments/lib$v.so musicstyletransfer_amd/csrc/libmst_hip.so
    for cfg in ${CFGS:-1 4}; do
      timeout -k 10 200 python bench.py --config $cfg --steps 60 > gpurun_out/ab_$v.json 2> gpurun_out/ab_$v.err || { echo "$v cfg$cfg FAILED" >> $OUT; exit 1; }
      python - "$v" "$cfg" "$rep" >> $OUT <<'PY'
import json,sys
v,cfg,rep=sys.argv[1:4]
d=json.loads(open("gpurun_out/ab_%s.json"%v).read().strip().splitlines()[-1])
att=" ".join("%s=%.1f"%(("bwd" if "bwd" in f["kernel"] else "fwd"), f["avg_launch_ms"]*1000) for f in d["roofline"]["families"] if "attention" in f["kernel"])
print("rep%s lib%s cfg%s ms=%.4f %s"%(rep,v,cfg,d["ms_per_step"],att))
PY
    done
  done
done
cat $OUT
