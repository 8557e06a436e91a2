mkdir -p gpurun_out/r3m
for lib in libmafed_hip lib_attn3 lib_attn4 lib_attn1; do
  echo "== $lib" >> gpurun_out/r3m/attn.log
  MAFED_HIP_LIB=$PWD/mafed_amd/$lib.so timeout -k 10 120 python tools/attn_bench.py >> gpurun_out/r3m/attn.log 2>&1 || exit 1
done
