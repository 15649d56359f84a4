#!/bin/bash
# A/B of the fused-attention backward generations under rocprofv3 --kernel-trace --stats (per-kernel average durations).
# usage: tools/flash_ab.sh <out_prefix> ; writes gpurun_out/<prefix>_{v1,v2}_{T}.csv summaries
set -e
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
for gen in v2 v1; do
  for shape in "32 256 64" "8 256 128"; do
    set -- $shape
    tag="${1}_${3}"
    if [ "$gen" = "v1" ]; then export ODVAE_FLASH_BWD_V1=1; else unset ODVAE_FLASH_BWD_V1; fi
    rm -rf /tmp/prof_$gen$tag
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$gen$tag -o run -- python3 $ROOT/tools/bf16_probe.py flash_bwd $1 $2 $3 3 > $OUT/$1_${gen}_$tag.log 2>&1 || true
    f=$(find /tmp/prof_$gen$tag -name "*kernel_stats.csv" | head -1)
    echo "== $gen N=$1 C=$2 HxW=$3x$3" >> $OUT/flash_ab_summary.txt
    if [ -n "$f" ]; then grep -i "flash" "$f" | cut -c1-220 >> $OUT/flash_ab_summary.txt; fi
  done
done
cat $OUT/flash_ab_summary.txt
