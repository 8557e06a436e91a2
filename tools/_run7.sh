set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4g
kg() { env TAG="$1-$2" MAFED_HIP_LIB=$GRAFT_REPO_ROOT/mafed_amd/lib_$1.so MAFED_HIP_LIB_LOOSE=1 GEMM_BENCH_PRE=$2 timeout -k 10 200 python tools/gemm_kernel_vs_gap.py 2>&1 | grep "loop" >> gpurun_out/r4g/kernel_vs_gap.txt; }
for r in 1 2; do kg r03 701; kg tk0 720; kg full 720; kg nopark 720; kg nohook 720; kg noboth 720; done
cat gpurun_out/r4g/kernel_vs_gap.txt
