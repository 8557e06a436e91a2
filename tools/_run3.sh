set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4c
timeout -k 10 900 python -m pytest tests/test_gpu_overwrite.py tests/test_gpu_oracle_fullshape.py tests/test_gpu_replay.py tests/test_gpu_ddp.py -x -q > gpurun_out/r4c/tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/r4c/tests.log
tail -15 gpurun_out/r4c/tests.log
B="--steps 20 --warmup 6 --no-secondary --no-teacher-cache-leg --no-image-leg --no-cpu-baseline --no-kernel-profile"
for i in 1 2; do
  python bench.py $B --no-ddp-forecast > gpurun_out/r4c/bench_ticket_$i.json 2> gpurun_out/r4c/bench_ticket_$i.log
  python bench.py $B --no-ddp-forecast --no-ticketed-order > gpurun_out/r4c/bench_static_$i.json 2> gpurun_out/r4c/bench_static_$i.log
done
python bench.py $B --no-ddp-forecast --hw-queues 8 > gpurun_out/r4c/bench_ticket_q8.json 2> gpurun_out/r4c/bench_ticket_q8.log
python bench.py $B --no-ddp-forecast --hw-queues 8 --no-ticketed-order > gpurun_out/r4c/bench_static_q8.json 2> gpurun_out/r4c/bench_static_q8.log
python bench.py $B > gpurun_out/r4c/bench_forecast.json 2> gpurun_out/r4c/bench_forecast.log
grep -h "timed region\|ddp forecast" gpurun_out/r4c/*.log
GEMM_BENCH_EPI=1 GEMM_BENCH_ONLY=qkv,dense,fc1,dfc2,dfc1,dqkv,dao timeout -k 10 300 python tools/gemm_bench.py 701 > gpurun_out/r4c/gemm_new.txt 2>&1
MAFED_HIP_LIB=$GRAFT_REPO_ROOT/mafed_amd/lib_r03.so MAFED_HIP_LIB_LOOSE=1 GEMM_BENCH_EPI=1 GEMM_BENCH_ONLY=qkv,dense,fc1,dfc2,dfc1,dqkv,dao timeout -k 10 300 python tools/gemm_bench.py 701 > gpurun_out/r4c/gemm_r03.txt 2>&1
GEMM_BENCH_EPI=1 GEMM_BENCH_ONLY=qkv,dense,fc1,dfc2,dfc1,dqkv,dao timeout -k 10 300 python tools/gemm_bench.py 701 > gpurun_out/r4c/gemm_new2.txt 2>&1
grep -h "NT\|NN" gpurun_out/r4c/gemm_new.txt gpurun_out/r4c/gemm_r03.txt gpurun_out/r4c/gemm_new2.txt
