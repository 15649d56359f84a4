#!/bin/bash
# timing-only ablations of the F(4x4) Winograd kernel (tools/ab_build.py w4abl<bits> conv3x3_wino4_f32.hip -DODVAE_W4_ABL=<bits>)
cd "$GRAFT_REPO_ROOT"
echo "== shipped"; WINO_NOCHECK=1 python tools/wino4_time.py
for v in 2 64 128 1024; do echo "== ABL $v"; WINO_NOCHECK=1 ODVAE_PROBE_LIB=$GRAFT_REPO_ROOT/tools/bin/libodvae_w4abl$v.so python tools/wino4_time.py; done
