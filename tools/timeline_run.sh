#!/bin/bash
# Run ON THE GPU BOX: kernel trace of the default bench step -> gpurun_out/tl/timeline.txt (tools/trace_timeline.py)
set -e
OUT=gpurun_out/tl
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_kt
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_kt -- python3 bench.py --no-cpu-baseline --no-kernel-profile --no-secondary --no-image-leg --no-teacher-cache-leg "$@" > $OUT/bench.json 2> $OUT/bench.err
kt=$(find gpurun_out/prof_kt -name "*kernel_trace.csv" | head -1)
python3 tools/trace_timeline.py "$kt" 8 > $OUT/timeline.txt
rm -rf gpurun_out/prof_kt
