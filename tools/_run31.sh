cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5o
timeout -k 10 200 python tools/decode_out_trace.py > gpurun_out/r5o/trace.txt 2>&1
grep -v amdgpu gpurun_out/r5o/trace.txt
timeout -k 10 300 python -m pytest tests/test_gpu_decode.py -x -q > gpurun_out/r5o/test.log 2>&1; tail -2 gpurun_out/r5o/test.log
timeout -k 10 300 python tools/decode_kernel_bench.py > gpurun_out/r5o/kernels.txt 2>&1; grep -v amdgpu gpurun_out/r5o/kernels.txt
timeout -k 10 300 python tools/decode_bench.py > gpurun_out/r5o/decode.txt 2>&1; grep -v amdgpu gpurun_out/r5o/decode.txt | cut -c1-250
