#!/bin/bash
# The data-parallel path (RCCL world size 1: bucketed GradReducer, pre-scaled loss) against the single-process step on one box, back to back,
# with the round-4 fusion switches off one at a time: does any of them interact with the reducer?  (No: 227.5 vs 228.0 ms, profiles/ / DESIGN.md 4.)
# usage (on the GPU box): bash tools/dp_ab.sh
run() { env $1 python bench.py $2 --steps 8 --warmup 3 --no-cpu-baseline --no-other-configs --no-kernel-events 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 $2', round(d['ms_per_step'],2), round(d['gpu_step_ms']['median'],2))"; }
run A=1 ""
run A=1 "--force-dist"
run ODVAE_GN_FUSED_BWD=0 "--force-dist"
run ODVAE_ATTN_FOLDED_SOFTMAX=0 "--force-dist"
run ODVAE_UPCONV_POOLED_DGRAD=0 "--force-dist"
run ODVAE_PACK_BATCH=0 "--force-dist"
run A=1 ""
