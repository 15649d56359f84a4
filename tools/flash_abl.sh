#!/bin/bash
# timing of the ablation builds of the dK/dV pair kernel (tools/bin/libodvae_abl*.so); prints rocprofv3 average kernel durations
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
for lib in "" $(ls $ROOT/tools/bin/libodvae_abl*.so 2>/dev/null); do
  if [ -n "$lib" ]; then export ODVAE_PROBE_LIB=$lib; else unset ODVAE_PROBE_LIB; fi
  rm -rf /tmp/prof_abl
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_abl -o run -- python3 $ROOT/tools/bf16_probe.py flash_bwd 8 256 128 3 > /tmp/abl.log 2>&1
  f=$(find /tmp/prof_abl -name "*kernel_stats.csv" | head -1)
  echo "== ${lib:-default}"
  grep -i "pair" "$f" | awk -F'","|",' '{print $1}' | cut -c1-60 | paste - <(grep -i "pair" "$f" | awk -F, '{print $(NF-5)}')
done
