#!/bin/bash
# Timing-only ablation builds of the Q15 MFCC kernel: one library per EQ_ABLATE mask under gpurun_out/abl/.
# usage (here): tools/ablate/q15.sh build "0 1 2 4 8 16 31"     (on the box): tools/ablate/q15.sh run "0 1 2 ..."
set -e
cd "$(dirname "$0")/../.."
mkdir -p edison_amd/csrc/abl
if [ "$1" = build ]; then
  for a in $2; do
    ED_CFLAGS="-DEQ_ABLATE=$a" python3 -m edison_amd.build --force > /dev/null 2>&1
    cp edison_amd/csrc/libedison_hip.so edison_amd/csrc/abl/libedison_hip_q$a.so
  done
  python3 -m edison_amd.build --force > /dev/null 2>&1
else
  for a in $2; do
    python3 tools/bench_q15.py --lib edison_amd/csrc/abl/libedison_hip_q$a.so --tag "abl$a"
  done
fi
