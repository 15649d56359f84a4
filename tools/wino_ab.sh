#!/bin/bash
# time tools/wino_time.py against a list of A/B libraries (tools/bin/libodvae_<name>.so; "base" = the shipped library)
cd "$GRAFT_REPO_ROOT" || exit 1
for v in "$@"; do
  if [ "$v" = base ]; then unset ODVAE_PROBE_LIB; else export ODVAE_PROBE_LIB=$GRAFT_REPO_ROOT/tools/bin/libodvae_$v.so; fi
  echo "== $v"
  timeout -k 10 120 python tools/wino_time.py 2>/dev/null || exit 1
done
