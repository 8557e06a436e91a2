#!/bin/bash
# Run ON THE GPU BOX from the repo root: L2-fabric-side bytes per launch of every bf16 GEMM shape of the 410M step, isolated launches,
# FETCH_SIZE and WRITE_SIZE in separate --pmc passes (MI355X_MICROARCH.md: they do not fit one pass) -> gpurun_out/pmc_shapes/*.csv,
# summarised by tools/pmc_traffic_summary.py
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_shapes
rm -rf $OUT && mkdir -p $OUT
for s in qkv dense fc1 fc2 dfc2 dfc1 dqkv dao wgrp2; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf gpurun_out/prof_tmp
    timeout -k 10 120 rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/prof_tmp -- python3 tools/gemm_shape_run.py $s 10 > /dev/null 2>&1
    f=$(find gpurun_out/prof_tmp -name "*counter_collection.csv" | head -1)
    cp "$f" $OUT/${s}_$c.csv
    echo "$s $c done"
  done
done
rm -rf gpurun_out/prof_tmp
python3 tools/pmc_traffic_summary.py $OUT > $OUT/pmc_traffic.json
