#!/bin/bash
# A/B of the -DODVAE_W4_OPT=<bits> builds of the F(4x4) kernel against the shipped library: correctness (13 shapes vs f64) then timing
cd "$GRAFT_REPO_ROOT"
echo "== shipped"; WINO_NOCHECK=1 python3 tools/wino4_time.py | grep "^B32"
for v in "$@"; do echo "== OPT $v"; ODVAE_PROBE_LIB=$GRAFT_REPO_ROOT/tools/bin/libodvae_w4opt$v.so python3 tools/wino4_time.py | grep "^B32\|WRONG\|Error\|error" ; done
echo "== shipped again"; WINO_NOCHECK=1 python3 tools/wino4_time.py | grep "^B32"
