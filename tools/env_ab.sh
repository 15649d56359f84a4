#!/bin/bash
# run one probe script under a list of environment settings: tools/env_ab.sh tools/gemm_probe.py "A=1" "A=2 B=3" ...
cd "$GRAFT_REPO_ROOT" || exit 1
script=$1; shift
for v in "$@"; do
  echo "== $v"
  env $v timeout -k 10 300 python $script 2>/dev/null || exit 1
done
