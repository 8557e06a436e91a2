set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4e
timeout -k 10 1000 python -m pytest tests/test_gpu_overwrite.py tests/test_gpu_oracle_fullshape.py tests/test_gpu_replay.py tests/test_gpu_ddp.py -x -q > gpurun_out/r4e/tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/r4e/tests.log
tail -12 gpurun_out/r4e/tests.log
gb() { env MAFED_HIP_LIB=$GRAFT_REPO_ROOT/mafed_amd/lib_$1.so MAFED_HIP_LIB_LOOSE=1 GEMM_BENCH_PRE=$2 GEMM_BENCH_EPI=1 GEMM_BENCH_ONLY=qkv,fc1,dfc2,dfc1,dao timeout -k 10 200 python tools/gemm_bench.py 701 2>&1 | grep "NT\|NN" | sed "s/^/$1 $2 /" >> gpurun_out/r4e/gemm_ab.txt; }
for r in 1 2; do gb r03 701; gb tk0 720; gb full 720; gb full 721; done
cat gpurun_out/r4e/gemm_ab.txt
B="--steps 20 --warmup 6 --no-secondary --no-teacher-cache-leg --no-image-leg --no-cpu-baseline --no-kernel-profile"
for i in 1 2; do
  (cd tools/_r03 && python bench.py $B > $GRAFT_REPO_ROOT/gpurun_out/r4e/bench_r03_$i.json 2> $GRAFT_REPO_ROOT/gpurun_out/r4e/bench_r03_$i.log)
  python bench.py $B --no-ddp-forecast > gpurun_out/r4e/bench_new_$i.json 2> gpurun_out/r4e/bench_new_$i.log
  MAFED_HIP_LIB=$GRAFT_REPO_ROOT/mafed_amd/lib_tk0.so python bench.py $B --no-ddp-forecast > gpurun_out/r4e/bench_tk0_$i.json 2> gpurun_out/r4e/bench_tk0_$i.log
done
grep -h "timed region" gpurun_out/r4e/bench_*.log
