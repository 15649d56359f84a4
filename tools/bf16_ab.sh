#!/bin/bash
# bf16 conv timings (three layer shapes) under a list of libraries: tools/bf16_ab.sh base head ...
cd "$GRAFT_REPO_ROOT" || exit 1
for v in "$@"; do
  if [ "$v" = base ]; then unset ODVAE_PROBE_LIB; else export ODVAE_PROBE_LIB=$GRAFT_REPO_ROOT/tools/bin/libodvae_$v.so; fi
  echo "== $v"
  for shape in "32 128 256" "32 256 64" "32 512 32"; do timeout -k 10 120 python tools/bf16_probe.py conv $shape 200 2>/dev/null || exit 1; done
done
