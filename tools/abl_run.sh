mkdir -p gpurun_out/r3o
timeout -k 10 600 python -m pytest tests/test_gpu_replay.py -m gpu -x -q > gpurun_out/r3o/tests.log 2>&1 || exit 1
timeout -k 10 400 python bench.py --no-secondary --no-image-leg --no-cpu-baseline > gpurun_out/r3o/bench.json 2> gpurun_out/r3o/bench.err
