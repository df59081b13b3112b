set -e
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py tests/test_gpu_stream.py -x -q > gpurun_out/r04_t2_pytest.log 2>&1 || { tail -30 gpurun_out/r04_t2_pytest.log; exit 1; }
tail -2 gpurun_out/r04_t2_pytest.log
python tools/lab/ab_cnn.py --rounds 10 cnnr03 prod 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_cnn_ab1.txt
