# the run that made profiles/r04_launch_len.txt: tools/lab/launch_len.py interleaved, then one rocprofv3 --kernel-trace --stats run per leg (GPU box, via gpurun)
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q > gpurun_out/r04_t0_pytest.log 2>&1 || { tail -30 gpurun_out/r04_t0_pytest.log; exit 1; }
tail -3 gpurun_out/r04_t0_pytest.log
python tools/lab/launch_len.py --rounds 8 > gpurun_out/r04_launch_len.txt 2>&1
cat gpurun_out/r04_launch_len.txt
for leg in cfg2:s cfg2:l kwsmix:s kwsmix:l; do
  d=gpurun_out/r04_ll_$(echo $leg | tr : _)
  rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 tools/lab/launch_len.py --rounds 4 --legs $leg > $d.txt 2>&1
  f=$(find $d -name "*kernel_stats.csv" | head -1)
  echo "== $leg"; grep ed_mfcc2 $f
done
