#!/bin/bash
# Build the library once per ED_CFLAGS variant (name=flags ...) for A/B timing of the Q15 kernel; run on the box.
# usage (here): tools/ablate/q15_flags.sh build base= w3=-DEQ_WAVES_PER_EU=3 ...   (box): tools/ablate/q15_flags.sh run base w3 ...
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
    python3 tools/bench_q15.py --lib edison_amd/csrc/abl/libedison_hip_$name.so --tag "$name" --check
  done
fi
