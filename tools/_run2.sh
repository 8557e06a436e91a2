set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4b
timeout -k 10 600 python -m pytest tests/test_gpu_gemm_pp.py -x -q > gpurun_out/r4b/test_pp.log 2>&1; echo "pytest rc $?" >> gpurun_out/r4b/test_pp.log
tail -5 gpurun_out/r4b/test_pp.log
grep -q "pytest rc 0" gpurun_out/r4b/test_pp.log || exit 1
timeout -k 10 300 python tools/contention_bench.py > gpurun_out/r4b/contention.txt 2>&1
cat gpurun_out/r4b/contention.txt
GEMM_BENCH_EPI=1 GEMM_BENCH_PRE=720 GEMM_BENCH_ONLY=qkv,dense,fc1,dfc2,dfc1,dqkv timeout -k 10 300 python tools/gemm_bench.py 701 > gpurun_out/r4b/gemm_bench_static.txt 2>&1
GEMM_BENCH_EPI=1 GEMM_BENCH_PRE=721 GEMM_BENCH_ONLY=qkv,dense,fc1,dfc2,dfc1,dqkv timeout -k 10 300 python tools/gemm_bench.py 701 > gpurun_out/r4b/gemm_bench_ticket.txt 2>&1
tail -8 gpurun_out/r4b/gemm_bench_static.txt gpurun_out/r4b/gemm_bench_ticket.txt
