cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4x
timeout -k 10 500 python bench.py > gpurun_out/r4x/bench_410m.json 2> gpurun_out/r4x/bench_410m.log; tail -c 600 gpurun_out/r4x/bench_410m.json | cut -c1-300
timeout -k 10 300 python bench.py --model 160m --no-cpu-baseline --no-secondary --no-image-leg --no-teacher-cache-leg --no-ddp-forecast > gpurun_out/r4x/bench_160m.json 2> gpurun_out/r4x/bench_160m.log; cut -c1-400 gpurun_out/r4x/bench_160m.json
timeout -k 10 400 python bench.py --model 1.4b --no-cpu-baseline --no-secondary --no-image-leg --no-teacher-cache-leg --no-ddp-forecast > gpurun_out/r4x/bench_1.4b.json 2> gpurun_out/r4x/bench_1.4b.log; cut -c1-400 gpurun_out/r4x/bench_1.4b.json
