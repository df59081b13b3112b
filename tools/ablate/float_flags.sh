#!/bin/bash
# Build the library once per ED_CFLAGS variant (name=flags ...) for A/B timing of the float MFCC kernel.
# usage (here): tools/ablate/float_flags.sh build base= t1=-DED_T1_LDS=1 ...   (box): tools/ablate/float_flags.sh run base t1 ...
set -e
cd "$(dirname "$0")/../.."
mkdir -p edison_amd/csrc/abl
mode=$1; shift
if [ "$mode" = build ]; then
  for kv in "$@"; do
    name=${kv%%=*}; flags=${kv#*=}
    ED_CFLAGS="$flags" python3 -m edison_amd.build --force > /dev/null 2>&1
    cp edison_amd/csrc/libedison_hip.so edison_amd/csrc/abl/libedison_hip_$name.so
  done
  python3 -m edison_amd.build --force > /dev/null 2>&1
else
  for name in "$@"; do
    EDISON_LIB=edison_amd/csrc/abl/libedison_hip_$name.so python3 tools/bench_mfcc.py --tag "$name" --frames ${FRAMES:-65536} --utts 64 --reps 50 2>&1 | grep "mfcc"
  done
fi
