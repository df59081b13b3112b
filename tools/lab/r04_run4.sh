set -e
cd $GRAFT_REPO_ROOT
for f in 65536 65536 1048576 8126464; do echo "=== frames $f, loop-only stamps, static slices"; python tools/lab/stamp_run.py --name stamp2 --cmax -1 --frames $f 2>&1 | grep -E "in-kernel clock|busy fraction|launch timeline|launch-to-launch"; done > gpurun_out/r04_clock_by_launch_len.txt
cat gpurun_out/r04_clock_by_launch_len.txt
