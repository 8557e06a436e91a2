set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4a
python tools/stream_cu_bench.py > gpurun_out/r4a/stream_cu.txt 2>&1
python tools/contention_bench.py > gpurun_out/r4a/contention_base.txt 2>&1
python bench.py --steps 20 --warmup 8 --no-secondary --no-teacher-cache-leg --no-image-leg --no-cpu-baseline > gpurun_out/r4a/bench.json 2> gpurun_out/r4a/bench.log
tail -3 gpurun_out/r4a/bench.log
