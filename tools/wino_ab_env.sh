#!/bin/bash
# time tools/wino_time.py under a list of environment settings of the shipped library, e.g. "ODVAE_WINO_PERSIST=0" "ODVAE_WINO_PERSIST=1"
cd "$GRAFT_REPO_ROOT" || exit 1
for v in "$@"; do
  echo "== $v"
  env $v timeout -k 10 300 python tools/wino_time.py 2>/dev/null || exit 1
done
