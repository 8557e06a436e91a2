set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4m
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r4m/gpu_tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/r4m/gpu_tests.log
tail -6 gpurun_out/r4m/gpu_tests.log
