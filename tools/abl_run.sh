mkdir -p gpurun_out/r3r
for k in 0 4 8 12 0 6; do
  echo "== MAFED_OLD_LAYERS=$k" >> gpurun_out/r3r/old.log
  MAFED_OLD_LAYERS=$k timeout -k 10 200 python bench.py --no-secondary --no-image-leg --no-cpu-baseline --no-kernel-profile --no-teacher-cache-leg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])" >> gpurun_out/r3r/old.log || exit 1
done
