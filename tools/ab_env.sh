#!/bin/bash
# A/B of one environment switch on the f32 (or --bf16) benchmark step, alternating runs in one box:
#   tools/ab_env.sh VAR "extra bench flags" [rounds]   -> prints ms_per_step per run
VAR=$1; FLAGS=$2; ROUNDS=${3:-2}
for r in $(seq $ROUNDS); do
  for v in 0 1; do
    env $VAR=$v python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-other-configs --no-kernel-events $FLAGS > /tmp/ab_$v.json 2>/tmp/ab_$v.err || { tail -5 /tmp/ab_$v.err; exit 1; }
    python - "$VAR" $v <<'PY'
import json, sys
d = json.loads(open("/tmp/ab_%s.json" % sys.argv[2]).read().strip().splitlines()[-1])
print("%s=%s %s images/s %.3f ms/step (median step %.3f)" % (sys.argv[1], sys.argv[2], d["value"], d["ms_per_step"], d.get("gpu_step_ms", {}).get("median", float("nan"))), flush=True)
PY
  done
done
