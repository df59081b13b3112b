#!/bin/bash
# Timing-only ablation builds of the MFCC kernel (results are WRONG by construction; only the time matters).
set -e
cd "$(dirname "$0")/../.."
cp edison_amd/csrc/mfcc_kernels.hip /tmp/mfcc_kernels.orig
for a in 0 1 2 3 4 8 12 15; do
  cp tools/ablate/mfcc_kernels_ablate.hip edison_amd/csrc/mfcc_kernels.hip
  ED_CFLAGS="-DED_ABLATE=$a" python3 -m edison_amd.build --force > /dev/null 2>&1
  python3 tools/bench_mfcc.py --tag abl$a --frames 262144 --utts 64 2>&1 | grep "mfcc B"
done
cp /tmp/mfcc_kernels.orig edison_amd/csrc/mfcc_kernels.hip
