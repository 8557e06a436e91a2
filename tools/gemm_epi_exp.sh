#!/bin/bash
# Run ON THE GPU BOX: A/B of GEMM builds (tools/build_variant.sh NAME ...) on the step's M = 9216 shapes with their real epilogues.
# usage: tools/gemm_epi_exp.sh NAME...   ("base" = mafed_amd/libmafed_hip.so)
cd "$(dirname "$0")/.."
for name in "$@"; do
  lib=mafed_amd/lib_$name.so
  [ "$name" = base ] && lib=mafed_amd/libmafed_hip.so
  echo "== $name"
  MAFED_HIP_LIB=$PWD/$lib GEMM_BENCH_EPI=1 GEMM_BENCH_ONLY="qkv,dense,fc1,fc2,dfc2,dfc1,wfc1,wdns" python3 tools/gemm_bench.py 0
done
