set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q > gpurun_out/r04_t1_pytest.log 2>&1 || { tail -30 gpurun_out/r04_t1_pytest.log; exit 1; }
tail -2 gpurun_out/r04_t1_pytest.log
python tools/lab/ab_mfcc.py --rounds 10 prod:cmax=-1 prod prod:cmax=8 prod:cmax=16 prod:cmax=24 prod:cmax=12,lead=10 prod:cmax=16,lead=12 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_tail_ab1.txt
