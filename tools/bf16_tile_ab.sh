#!/bin/bash
# step-level A/B of the bf16 conv tile selection (ODVAE_CONV_BF16_WIDE2): bf16 256x256 B=32 and configs[4] (512x512, checkpointed Decoder)
cd "$GRAFT_REPO_ROOT"
for v in "$@"; do
  ODVAE_CONV_BF16_WIDE2=$v python bench.py --bf16 --steps 15 --warmup 5 --no-other-configs --no-cpu-baseline --no-kernel-events 2>/dev/null > /tmp/ab_$v.json
  python -c "import json; d=json.load(open('/tmp/ab_$v.json')); print('tile $v  256x256: %.1f images/s  %.2f ms' % (d['value'], d['ms_per_step']))"
  ODVAE_CONV_BF16_WIDE2=$v python bench.py --bf16 --res 512 --batch 32 --ckpt-decoder --steps 4 --warmup 2 --no-other-configs --no-cpu-baseline --no-kernel-events 2>/dev/null > /tmp/ab5_$v.json
  python -c "import json; d=json.load(open('/tmp/ab5_$v.json')); print('tile $v  512x512: %.1f images/s  %.2f ms' % (d['value'], d['ms_per_step']))"
done
