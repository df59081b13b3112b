set -e
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q > gpurun_out/r04_t3_pytest.log 2>&1 || { tail -30 gpurun_out/r04_t3_pytest.log; exit 1; }
tail -2 gpurun_out/r04_t3_pytest.log
