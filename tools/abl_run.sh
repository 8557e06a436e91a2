mkdir -p gpurun_out/r3j
timeout -k 10 900 python -m pytest tests/test_gpu_ddp.py -m gpu -x -q > gpurun_out/r3j/tests.log 2>&1 || exit 1
MAFED_DIST_BACKEND=gloo timeout -k 10 300 python bench.py --gpus 2 --steps 5 --warmup 2 --no-secondary --no-image-leg --no-kernel-profile --exact-normaliser --memory-size 1000 > gpurun_out/r3j/bench2.json 2> gpurun_out/r3j/bench2.err || exit 1
timeout -k 10 300 python bench.py --no-secondary --no-image-leg --no-cpu-baseline > gpurun_out/r3j/bench1.json 2> gpurun_out/r3j/bench1.err
