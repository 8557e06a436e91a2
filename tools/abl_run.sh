mkdir -p gpurun_out/r3i
timeout -k 10 300 python -m pytest tests/test_gpu_gemm_pp.py -m gpu -x -q > gpurun_out/r3i/pp_tests.log 2>&1 || exit 1
for lib in lib_pd3 libmafed_hip lib_pd12 lib_pd3 libmafed_hip lib_pd12; do
    echo "== $lib" >> gpurun_out/r3i/pd.log
    MAFED_HIP_LIB=$PWD/mafed_amd/$lib.so GEMM_BENCH_EPI=1 GEMM_BENCH_ONLY=fc1,fc2,dfc2,qkv timeout -k 10 120 python tools/gemm_bench.py 700,701 >> gpurun_out/r3i/pd.log 2>&1 || exit 1
done
