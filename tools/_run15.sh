set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4q
timeout -k 10 600 python -m pytest tests/test_gpu_decode.py -x -q > gpurun_out/r4q/test_decode.log 2>&1; echo "pytest rc $?" >> gpurun_out/r4q/test_decode.log
tail -4 gpurun_out/r4q/test_decode.log
grep -q "pytest rc 0" gpurun_out/r4q/test_decode.log || exit 1
timeout -k 10 300 python tools/decode_bench.py > gpurun_out/r4q/decode.txt 2>&1
cat gpurun_out/r4q/decode.txt
SKIP_PMC=1 SKIP_BENCH=1 bash tools/make_profiles.sh > gpurun_out/r4q/prof.log 2>&1
tail -3 gpurun_out/r4q/prof.log
