#!/bin/bash
# usage (on the GPU box, from the repo root): [MST_PROFILE_CONFIG=N] tools/profile_round.sh <tag> [bench args... e.g. --config N]
# rocprofv3 kernel stats of `python3 bench.py <args>` plus the two PMC passes for HBM traffic (separate --pmc runs, as
# MI355X_MICROARCH.md's HBM section prescribes), then tools/make_profile_summary.py -> profiles/<tag>_* and
# gpurun_out/<tag>/profiles/ (copied back by gpurun).
set -e
tag="$1"; shift
root="${GRAFT_REPO_ROOT:-$(pwd)}"
out="$root/gpurun_out/$tag"
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -o r -- python3 "$root/bench.py" --steps 60 --warmup 10 --no-cpu-baseline "$@" > "$out/bench.json" 2> "$out/bench.err"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out/fetch" -o r -- python3 "$root/bench.py" --steps 3 --warmup 1 --no-cpu-baseline "$@" > /dev/null 2> "$out/fetch.err"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$out/write" -o r -- python3 "$root/bench.py" --steps 3 --warmup 1 --no-cpu-baseline "$@" > /dev/null 2> "$out/write.err"
cd "$root"
python3 tools/make_profile_summary.py "$tag" "$out/stats" "$out/fetch" "$out/write" "$out/bench.json"
mkdir -p "$out/profiles" && cp profiles/${tag}_* profiles/roofline_traffic.json "$out/profiles/"
python3 tools/step_timeline.py "$out/stats/r_kernel_trace.csv" > "$out/profiles/${tag}_timeline.txt"
