cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/full
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/full/test_gpu.log 2>&1; echo "pytest rc $?" >> gpurun_out/full/test_gpu.log
tail -6 gpurun_out/full/test_gpu.log | cut -c1-300
