mkdir -p gpurun_out/r3g
for lib in libmafed_hip lib_abl5 lib_abl6 lib_abl7; do
  for epi in 0 1; do
    echo "== $lib EPI=$epi" >> gpurun_out/r3g/abl.log
    MAFED_HIP_LIB=$PWD/mafed_amd/$lib.so GEMM_BENCH_NOCHECK=1 GEMM_BENCH_EPI=$epi GEMM_BENCH_ONLY=qkv,dense,fc1,dfc2,dqkv timeout -k 10 120 python tools/gemm_bench.py 710 >> gpurun_out/r3g/abl.log 2>&1 || exit 1
  done
done
