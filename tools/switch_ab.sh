#!/bin/bash
# Every A/B switch of the f32 step, one at a time, on ONE box: ms per step with the switch off against the default build's
# (12 timed steps after 4 warm-up steps each; the default is run first and last).  usage (GPU box): bash tools/switch_ab.sh
run() { env $1 python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-other-configs --no-kernel-events 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-36s %8.2f ms/step  %7.2f images/s' % ('$1', d['ms_per_step'], d['value']))"; }
run DEFAULT=1
for s in ODVAE_GN_FUSED_BWD=0 ODVAE_UPCONV_POOLED_DGRAD=0 ODVAE_ATTN_FOLDED_SOFTMAX=0 ODVAE_PACK_BATCH=0 ODVAE_GN_FUSED_STATS=0 ODVAE_UPCONV_WINOGRAD4=0 ODVAE_CONV_WINOGRAD4=0 ODVAE_WGRAD_WINOGRAD=0 ODVAE_CONV_WINOGRAD=0 ODVAE_ATTN_UNFUSED=1 ODVAE_WINO_PERSIST=0; do run $s; done
run DEFAULT=1
