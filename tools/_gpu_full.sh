cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/full3
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/full3/test_gpu.log 2>&1; echo "pytest rc $?" >> gpurun_out/full3/test_gpu.log
tail -4 gpurun_out/full3/test_gpu.log | cut -c1-300
grep -q "pytest rc 0" gpurun_out/full3/test_gpu.log || exit 1
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/full3/smoke.log 2>&1; tail -1 gpurun_out/full3/smoke.log
