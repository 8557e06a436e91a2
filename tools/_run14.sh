set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4n
kg() { env TAG="$1-$2" MAFED_HIP_LIB=$GRAFT_REPO_ROOT/mafed_amd/lib_$1.so MAFED_HIP_LIB_LOOSE=1 GEMM_BENCH_PRE=$2 timeout -k 10 200 python tools/gemm_kernel_vs_gap.py 2>&1 | grep "loop" >> gpurun_out/r4n/kernel_vs_gap.txt; }
for r in 1 2; do kg r03 701; kg tk2 722; kg tk3 722; kg tk4 722; done
cat gpurun_out/r4n/kernel_vs_gap.txt
B="--steps 20 --warmup 6 --no-secondary --no-teacher-cache-leg --no-image-leg --no-cpu-baseline --no-kernel-profile --no-ddp-forecast"
for i in 1 2 3; do
  python bench.py $B > gpurun_out/r4n/bench_auto_$i.json 2> gpurun_out/r4n/bench_auto_$i.log
  python bench.py $B --gemm-variant 731 > gpurun_out/r4n/bench_fc2pp_$i.json 2> gpurun_out/r4n/bench_fc2pp_$i.log
  (cd tools/_r03 && python bench.py --steps 20 --warmup 6 --no-secondary --no-teacher-cache-leg --no-image-leg --no-cpu-baseline --no-kernel-profile > $GRAFT_REPO_ROOT/gpurun_out/r4n/bench_r03_$i.json 2> $GRAFT_REPO_ROOT/gpurun_out/r4n/bench_r03_$i.log)
done
grep -h "timed region" gpurun_out/r4n/bench_*.log
timeout -k 10 600 python -m pytest tests/test_gpu_oracle_fullshape.py -x -q > gpurun_out/r4n/t.log 2>&1; tail -2 gpurun_out/r4n/t.log
