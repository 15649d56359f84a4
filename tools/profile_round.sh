#!/bin/bash
# Round profile set on the GPU box: kernel-trace summaries of the fp32 headline step and the bf16 steps, plus the HBM-traffic PMC
# passes (FETCH_SIZE / WRITE_SIZE, separate passes) of the dominant conv kernels.  Leaves small files under gpurun_out/ only.
# usage: tools/profile_round.sh <round tag, e.g. r02>
set -e
tag=${1:-r02}
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
tools/profile_step.sh ${tag}_f32 "B=32, 256x256, fp32, rec+KL only" --steps 3 --warmup 1
tools/profile_step.sh ${tag}_bf16_256 "B=32, 256x256, bf16 mixed precision, rec+KL only" --bf16 --steps 3 --warmup 1
tools/profile_step.sh ${tag}_bf16_512 "B=32, 512x512, z=32x32x16, bf16 mixed precision, activation-checkpointed Decoder" --bf16 --res 512 --batch 32 --ckpt-decoder --steps 2 --warmup 1
cd /tmp && export TMPDIR=/tmp
for cfg in "f32:conv3x3_wino4_kernel:" "bf16:conv_bf16_kernel:--bf16"; do
  name=${cfg%%:*}; rest=${cfg#*:}; kern=${rest%%:*}; flag=${rest#*:}
  for ctr in FETCH_SIZE WRITE_SIZE; do
    out=/tmp/pmc_${name}_$ctr; rm -rf $out
    rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $out -o p -- python3 $root/bench.py $flag --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-events --no-other-configs > /dev/null 2> /tmp/pmc_${name}_$ctr.err
  done
  python3 $root/tools/collect_traffic.py /tmp/pmc_${name}_FETCH_SIZE /tmp/pmc_${name}_WRITE_SIZE $root/gpurun_out/${tag}_${name}_conv_traffic.json $kern
  rm -rf /tmp/pmc_${name}_FETCH_SIZE /tmp/pmc_${name}_WRITE_SIZE
done
