cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5f
timeout -k 10 240 python -m pytest tests/test_gpu_decode.py -x -q -k "flow or one_launch" > gpurun_out/r5f/test.log 2>&1; tail -3 gpurun_out/r5f/test.log | cut -c1-250
grep -q "failed\|error" gpurun_out/r5f/test.log && exit 1
timeout -k 10 300 python tools/decode_bench.py > gpurun_out/r5f/decode.txt 2>&1
grep "decode step" gpurun_out/r5f/decode.txt
