set -e
mkdir -p gpurun_out/r3h
bash tools/pmc_traffic.sh > gpurun_out/r3h/pmc.log 2>&1
cp gpurun_out/pmc_shapes/pmc_traffic.json gpurun_out/r3h/
( echo "# tools/gemm_bench.py 700,701 with the step's epilogues (GEMM_BENCH_EPI=1): v700 = round-2 kernels (gemm_bf16_glds_kernel), v701 = automatic dispatch (persistent gemm_pp_kernel / gemm_z_kernel where a shape tiles them)"; GEMM_BENCH_EPI=1 timeout -k 10 200 python tools/gemm_bench.py 700,701 ) > gpurun_out/r3h/gemm_isolated.txt 2>&1
( echo "# same, plain products (no bias / activation / residual work in the epilogue)"; timeout -k 10 200 python tools/gemm_bench.py 700,701 ) >> gpurun_out/r3h/gemm_isolated.txt 2>&1
for L in 1 2 4; do timeout -k 10 100 python tools/gemm_group_bench.py $L >> gpurun_out/r3h/gemm_isolated.txt 2>&1; done
