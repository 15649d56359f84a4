#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for v in 2048 2175; do echo "== ABL $v"; WINO_NOCHECK=1 ODVAE_PROBE_LIB=$GRAFT_REPO_ROOT/tools/bin/libodvae_w4abl$v.so python3 tools/wino4_time.py | grep "^B32"; done
