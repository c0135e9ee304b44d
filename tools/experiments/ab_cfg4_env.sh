#!/bin/bash
# in-call A/B at configs[4]: tools/experiments/ab_cfg4_env.sh VAR=VALUE  -> ms per step and the attention backward family
run() { echo -n "$1: "; env $1 python bench.py --config 4 --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); f={x['kernel'][:40]: round(x['avg_launch_ms']*1e3,1) for x in d['roofline']['families'] if x['kernel'].startswith('attention bwd')}
print('ms/step', round(d['ms_per_step'],4), f)"; }
run A=0; run "$1"; run A=0; run "$1"
