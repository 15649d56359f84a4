#!/bin/bash
# per-layer A/B of conv_bf16_kernel on one box: default library against an A/B build (python tools/ab_build.py <name> conv_bf16.hip -D...)
# usage (GPU box): bash tools/convb_ab.sh tools/bin/libodvae_<name>.so
alt=$1
for shape in "32 128 256" "32 256 128" "32 512 64" "32 128 128" "32 512 32"; do
  for lib in "" "$alt" "" "$alt"; do
    echo -n "[${lib:-default}] "; ODVAE_PROBE_LIB=$lib python tools/bf16_probe.py conv $shape 20 2>/dev/null
  done
done
