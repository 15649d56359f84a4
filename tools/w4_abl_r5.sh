#!/bin/bash
# round 5: combined timing-only ablations of the F(4x4) kernel (bits: tools/w4_abl.sh / conv3x3_wino4_f32.hip)
cd "$GRAFT_REPO_ROOT"
echo "== shipped"; WINO_NOCHECK=1 python3 tools/wino4_time.py | grep "^B32"
for v in 1 66 67 52 60 127; do echo "== ABL $v"; WINO_NOCHECK=1 ODVAE_PROBE_LIB=$GRAFT_REPO_ROOT/tools/bin/libodvae_w4abl$v.so python3 tools/wino4_time.py | grep "^B32"; done
