#!/bin/bash
# f32 headline only: kernel-trace summary of the last step + the HBM-traffic PMC passes of the dominant kernel (see profile_round.sh)
set -e
tag=${1:-r03b}
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
tools/profile_step.sh ${tag}_f32 "B=32, 256x256, fp32, rec+KL only" --steps 3 --warmup 1
cd /tmp && export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  out=/tmp/pmc_f32_$ctr; rm -rf $out
  rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $out -o p -- python3 $root/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-events --no-other-configs > /dev/null 2> /tmp/pmc_f32_$ctr.err
done
python3 $root/tools/collect_traffic.py /tmp/pmc_f32_FETCH_SIZE /tmp/pmc_f32_WRITE_SIZE $root/gpurun_out/${tag}_f32_conv_traffic.json conv3x3_wino4_kernel
rm -rf /tmp/pmc_f32_FETCH_SIZE /tmp/pmc_f32_WRITE_SIZE
