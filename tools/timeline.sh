#!/bin/bash
# usage (GPU box, repo root): tools/timeline.sh <tag> [bench args]: rocprofv3 kernel trace of bench.py -> gpurun_out/<tag>/timeline.txt (one replayed step)
tag="$1"; shift
root="${GRAFT_REPO_ROOT:-$(pwd)}"; out="$root/gpurun_out/$tag"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -o r -- python3 "$root/bench.py" --steps 60 --warmup 10 --no-cpu-baseline "$@" > "$out/bench.json" 2> "$out/bench.err"
cd "$root" && python3 tools/step_timeline.py "$out/stats/r_kernel_trace.csv" > "$out/timeline.txt"
