cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5a
timeout -k 10 300 python -m pytest tests/test_gpu_overwrite.py -x -q > gpurun_out/r5a/test.log 2>&1; tail -4 gpurun_out/r5a/test.log | cut -c1-300
grep -q "failed" gpurun_out/r5a/test.log && exit 1
timeout -k 10 300 python bench.py --model 160m --no-cpu-baseline --no-secondary --no-image-leg --no-teacher-cache-leg --no-ddp-forecast > gpurun_out/r5a/bench_160m.json 2> gpurun_out/r5a/bench_160m.log; cut -c1-300 gpurun_out/r5a/bench_160m.json
timeout -k 10 400 python bench.py --model 1.4b --no-cpu-baseline --no-secondary --no-image-leg --no-teacher-cache-leg --no-ddp-forecast > gpurun_out/r5a/bench_1.4b.json 2> gpurun_out/r5a/bench_1.4b.log; cut -c1-300 gpurun_out/r5a/bench_1.4b.json
