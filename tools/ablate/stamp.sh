#!/bin/bash
# build the stamped diagnostic kernel, run it, restore the real kernel (run on the GPU box)
set -e
cd "$(dirname "$0")/../.."
python3 tools/ablate/make_stamp.py
cp edison_amd/csrc/mfcc_kernels.hip /tmp/mfcc_kernels.orig
cp tools/ablate/mfcc_kernels_stamp.hip edison_amd/csrc/mfcc_kernels.hip
python3 -m edison_amd.build --force > /dev/null 2>&1 || { cp /tmp/mfcc_kernels.orig edison_amd/csrc/mfcc_kernels.hip; exit 1; }
python3 tools/ablate/stamp_run.py ${1:-262144} 2>&1 | grep -v amdgpu
cp /tmp/mfcc_kernels.orig edison_amd/csrc/mfcc_kernels.hip
