#!/bin/bash
# the F(4x4)-domain weight-gradient probe against the library's F(2x2)-domain kernel on the step's layer shapes (B = 32)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for s in "32 256 256 128 128 128 256 512" "32 128 128 128 128 64 128 256" "32 128 128 256 256 64 128" "32 64 64 256 256 32 64 128" "32 64 64 512 512 16 32 64" "32 32 32 512 512 8 16 32" "32 32 32 256 512 8 16"; do
  tools/bin/wgrad_wino4_next $s | grep splits
done
python3 tools/wgrad_wino_probe.py time
