# usage: tools/experiments/trace_step.sh <tag> : kernel trace of a short bench run -> gpurun_out/<tag>_timeline.txt
tag="$1"; root="${GRAFT_REPO_ROOT:-$(pwd)}"; out="$root/gpurun_out/$tag"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$out/stats" -o r -- python3 "$root/bench.py" --steps 40 --warmup 10 --no-cpu-baseline > "$out/bench.json" 2> "$out/bench.err"
cd "$root"
python3 tools/step_timeline.py "$out/stats/r_kernel_trace.csv" > "$root/gpurun_out/${tag}_timeline.txt"
