mkdir -p gpurun_out/r3t
timeout -k 10 300 python -m pytest tests/test_gpu_gemm_pp.py -m gpu -x -q > gpurun_out/r3t/pp_tests.log 2>&1 || exit 1
for rep in 1 2 3; do
for lib in lib_cap256 libmafed_hip; do
  echo "== $lib" >> gpurun_out/r3t/ab.log
  MAFED_HIP_LIB=$PWD/mafed_amd/$lib.so timeout -k 10 200 python bench.py --no-secondary --no-image-leg --no-cpu-baseline --no-kernel-profile --no-teacher-cache-leg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])" >> gpurun_out/r3t/ab.log || exit 1
done
done
