#!/bin/bash
# rocprofv3 kernel trace of bench.py on the GPU box; leaves only the per-kernel summary of the last step and the stats CSV under
# gpurun_out/ (the raw trace is tens of MB).   usage: tools/profile_step.sh <tag> "<description>" <bench.py args...>
set -e
tag=$1; desc=$2; shift 2
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=/tmp/prof_$tag
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o $tag -- python3 $root/bench.py "$@" --no-cpu-baseline --no-kernel-events --no-other-configs > $root/gpurun_out/prof_$tag.json 2> $root/gpurun_out/prof_$tag.err
trace=$(find $out -name "*kernel_trace.csv" | head -1)
stats=$(find $out -name "*kernel_stats.csv" | head -1)
python3 $root/tools/summarize_trace.py $trace "$desc" > $root/gpurun_out/prof_${tag}_step_summary.md
cp $stats $root/gpurun_out/prof_${tag}_kernel_stats.csv
rm -rf $out
