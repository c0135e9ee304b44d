cfg=$1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/cfg$cfg/stats -o r -- python3 $GRAFT_REPO_ROOT/bench.py --config $cfg --steps 40 --warmup 10 --no-cpu-baseline > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/step_timeline.py gpurun_out/cfg$cfg/stats/r_kernel_trace.csv > gpurun_out/cfg$cfg/timeline.txt
cat gpurun_out/cfg$cfg/timeline.txt
