set -e
cd $GRAFT_REPO_ROOT
for c in -1 0 16; do echo "=== cmax $c, full stamps"; python tools/lab/stamp_run.py --name stamp --cmax $c 2>&1 | grep -v amdgpu.ids | head -24; done > gpurun_out/r04_stamps_tail.txt
for c in -1 0 16; do echo "=== cmax $c, loop-only stamps"; python tools/lab/stamp_run.py --name stamp2 --cmax $c 2>&1 | grep -v amdgpu.ids | head -24; done >> gpurun_out/r04_stamps_tail.txt
cat gpurun_out/r04_stamps_tail.txt
