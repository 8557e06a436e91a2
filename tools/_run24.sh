cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5d
timeout -k 10 240 python -m pytest tests/test_gpu_decode.py -x -q -k "flow or one_launch" > gpurun_out/r5d/test.log 2>&1; tail -30 gpurun_out/r5d/test.log | cut -c1-250
