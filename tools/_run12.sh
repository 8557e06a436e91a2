set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4l
timeout -k 10 900 python -m pytest tests/test_gpu_fullsize.py -x -q -k contended > gpurun_out/r4l/test_pp.log 2>&1; echo "pytest rc $?" >> gpurun_out/r4l/test_pp.log
tail -3 gpurun_out/r4l/test_pp.log
grep -q "pytest rc 0" gpurun_out/r4l/test_pp.log || exit 1
kg() { env TAG="$1-$2" MAFED_HIP_LIB=$GRAFT_REPO_ROOT/mafed_amd/lib_$1.so MAFED_HIP_LIB_LOOSE=1 GEMM_BENCH_PRE=$2 timeout -k 10 200 python tools/gemm_kernel_vs_gap.py 2>&1 | grep "loop" >> gpurun_out/r4l/kernel_vs_gap.txt; }
for r in 1 2; do kg r03 701; kg tk2 722; kg tk2 721; done
cat gpurun_out/r4l/kernel_vs_gap.txt
B="--steps 20 --warmup 6 --no-secondary --no-teacher-cache-leg --no-image-leg --no-cpu-baseline --no-kernel-profile"
for i in 1 2 3; do
  (cd tools/_r03 && python bench.py $B > $GRAFT_REPO_ROOT/gpurun_out/r4l/bench_r03_$i.json 2> $GRAFT_REPO_ROOT/gpurun_out/r4l/bench_r03_$i.log)
  python bench.py $B --no-ddp-forecast > gpurun_out/r4l/bench_auto_$i.json 2> gpurun_out/r4l/bench_auto_$i.log
  python bench.py $B --no-ddp-forecast --tile-order ticketed > gpurun_out/r4l/bench_ticket_$i.json 2> gpurun_out/r4l/bench_ticket_$i.log
done
python bench.py $B > gpurun_out/r4l/bench_forecast.json 2> gpurun_out/r4l/bench_forecast.log
grep -h "timed region\|ddp forecast" gpurun_out/r4l/bench_*.log
