#!/bin/bash
# kernel trace of the data-parallel step on ONE GPU (RCCL world size 1) -> gpurun_out/r05_dp_timeline.md
set -e
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=/tmp/prof_dp; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $out -o dp -- python3 $root/bench.py --force-dist --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-events --no-other-configs > $root/gpurun_out/prof_dp.json 2> $root/gpurun_out/prof_dp.err
python3 $root/tools/dp_timeline.py $(find $out -name "*kernel_trace.csv" | head -1) > $root/gpurun_out/r05_dp_timeline.md
rm -rf $out
