set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4d
timeout -k 10 900 python -m pytest tests/test_gpu_overwrite.py tests/test_gpu_oracle_fullshape.py tests/test_gpu_replay.py tests/test_gpu_ddp.py -x -q > gpurun_out/r4d/tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/r4d/tests.log
tail -8 gpurun_out/r4d/tests.log
E="GEMM_BENCH_EPI=1 GEMM_BENCH_ONLY=qkv,fc1,dfc2,dense"
for v in cap256 cap252 r03 cap256 cap252 r03; do
  env MAFED_HIP_LIB=$GRAFT_REPO_ROOT/mafed_amd/lib_$v.so MAFED_HIP_LIB_LOOSE=1 GEMM_BENCH_EPI=1 GEMM_BENCH_ONLY=qkv,fc1,dfc2,dense timeout -k 10 200 python tools/gemm_bench.py 701 2>&1 | grep "NT\|NN" | sed "s/^/$v ticket /" >> gpurun_out/r4d/gemm_ab.txt
done
for v in cap256 cap252; do
  env MAFED_HIP_LIB=$GRAFT_REPO_ROOT/mafed_amd/lib_$v.so GEMM_BENCH_PRE=720 GEMM_BENCH_EPI=1 GEMM_BENCH_ONLY=qkv,fc1,dfc2,dense timeout -k 10 200 python tools/gemm_bench.py 701 2>&1 | grep "NT\|NN" | sed "s/^/$v static /" >> gpurun_out/r4d/gemm_ab.txt
done
cat gpurun_out/r4d/gemm_ab.txt
B="--steps 20 --warmup 6 --no-secondary --no-teacher-cache-leg --no-image-leg --no-cpu-baseline --no-kernel-profile"
for i in 1 2; do
  (cd tools/_r03 && python bench.py $B > $GRAFT_REPO_ROOT/gpurun_out/r4d/bench_r03_$i.json 2> $GRAFT_REPO_ROOT/gpurun_out/r4d/bench_r03_$i.log)
  MAFED_HIP_LIB=$GRAFT_REPO_ROOT/mafed_amd/lib_cap256.so python bench.py $B --no-ddp-forecast > gpurun_out/r4d/bench_cap256_$i.json 2> gpurun_out/r4d/bench_cap256_$i.log
  MAFED_HIP_LIB=$GRAFT_REPO_ROOT/mafed_amd/lib_cap252.so python bench.py $B --no-ddp-forecast > gpurun_out/r4d/bench_cap252_$i.json 2> gpurun_out/r4d/bench_cap252_$i.log
done
grep -h "timed region" gpurun_out/r4d/bench_*.log
