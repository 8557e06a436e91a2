mkdir -p gpurun_out/r3n
for nc in 0 1 0 1; do
echo "== NOCOLSUM=$nc" >> gpurun_out/r3n/dfc2.log
GEMM_BENCH_NOCOLSUM=$nc GEMM_BENCH_EPI=1 GEMM_BENCH_ONLY=dfc2 timeout -k 10 120 python tools/gemm_bench.py 700,701 >> gpurun_out/r3n/dfc2.log 2>&1 || exit 1
done
GEMM_BENCH_ONLY=dfc2 timeout -k 10 120 python tools/gemm_bench.py 700,701 >> gpurun_out/r3n/dfc2.log 2>&1
