#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root: regenerates the judged artifacts under gpurun_out/profiles_new/
#   bench_line.json            the default bench.py line (roofline from live HIP events, cpu_baseline)
#   bench_kernel_stats.csv     rocprofv3 --kernel-trace --stats summary of the same command (without the CPU leg)
#   bench_kernel_by_shape.csv  per (kernel, grid) launch statistics from the same trace
#   timeline.txt               tools/trace_timeline.py on the same trace (overlap depth, per-queue mix, idle gaps)
#   pmc_traffic.json           fabric-side bytes per GEMM launch, per shape, from FETCH_SIZE / WRITE_SIZE (separate --pmc passes)
#   decode.txt, decode_kernel_stats.csv, decode_pmc.json, decode_flow_trace.txt   the validation decode (skip with SKIP_DECODE=1)
set -e
OUT=gpurun_out/profiles_new
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
if [ "$SKIP_BENCH" != "1" ]; then timeout -k 10 500 python3 bench.py --dump-profile $OUT 2> $OUT/bench_stderr.log | tail -1 > $OUT/bench_line.json; fi
rm -rf gpurun_out/prof_kt
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_kt -- python3 bench.py --no-cpu-baseline --no-kernel-profile --no-secondary --no-image-leg --no-teacher-cache-leg --no-ddp-forecast > /dev/null 2>&1
ks=$(find gpurun_out/prof_kt -name "*kernel_stats.csv" | head -1)
kt=$(find gpurun_out/prof_kt -name "*kernel_trace.csv" | head -1)
python3 - "$ks" "$kt" $OUT <<'PY'
import csv, re, sys
from collections import defaultdict
ks, kt, out = sys.argv[1:4]
short = lambda n: re.sub(r"\(.*", "", n).replace("void mafed::", "").replace("mafed::", "")
with open(ks) as f, open(out + "/bench_kernel_stats.csv", "w") as g:
    g.write("# rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --no-cpu-baseline --no-kernel-profile --no-secondary --no-image-leg --no-teacher-cache-leg --no-ddp-forecast   (1x MI355X)\n")
    r = csv.DictReader(f)
    g.write("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs\n")
    for row in r:
        g.write('"%s",%s,%s,%d,%s,%s,%s\n' % (short(row["Name"]), row["Calls"], row["TotalDurationNs"], float(row["AverageNs"]),
                                              row["Percentage"], row["MinNs"], row["MaxNs"]))
agg = defaultdict(lambda: [0, 0])
with open(kt) as f:
    for row in csv.DictReader(f):
        k = (short(row["Kernel_Name"]), int(row["Grid_Size_X"]) // max(1, int(row["Workgroup_Size_X"])))
        agg[k][0] += 1
        agg[k][1] += int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
with open(out + "/bench_kernel_by_shape.csv", "w") as g:
    g.write("# per (kernel, workgroups in x) launch statistics from the kernel trace of the same run (durations overlap across streams)\n")
    g.write("kernel,workgroups,launches,avg_us,total_ms\n")
    for (n, wg), (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:60]:
        g.write('"%s",%d,%d,%.1f,%.2f\n' % (n, wg, c, t / c / 1e3, t / 1e6))
PY
python3 tools/trace_timeline.py "$kt" 8 > $OUT/timeline.txt
rm -rf gpurun_out/prof_kt
# the same with every kernel alone on the chip (--no-overlap: one stream): what the per-kernel `*_no_overlap` figures of the line are checked against
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_kt -- python3 bench.py --no-cpu-baseline --no-kernel-profile --no-secondary --no-image-leg --no-teacher-cache-leg --no-ddp-forecast --no-overlap --no-pipeline-optimizer > /dev/null 2>&1
ks=$(find gpurun_out/prof_kt -name "*kernel_stats.csv" | head -1)
python3 - "$ks" $OUT <<'PY'
import csv, re, sys
ks, out = sys.argv[1:3]
short = lambda n: re.sub(r"\(.*", "", n).replace("void mafed::", "").replace("mafed::", "")
with open(ks) as f, open(out + "/bench_no_overlap_kernel_stats.csv", "w") as g:
    g.write("# rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --no-cpu-baseline --no-kernel-profile --no-secondary --no-image-leg --no-teacher-cache-leg --no-ddp-forecast --no-overlap --no-pipeline-optimizer   (1x MI355X; one stream)\n")
    g.write("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs\n")
    for row in csv.DictReader(f):
        g.write('"%s",%s,%s,%d,%s,%s,%s\n' % (short(row["Name"]), row["Calls"], row["TotalDurationNs"], float(row["AverageNs"]), row["Percentage"], row["MinNs"], row["MaxNs"]))
PY
rm -rf gpurun_out/prof_kt
python3 tools/step_timeline.py > $OUT/step_timeline_inlib.txt 2>/dev/null || true
if [ "$SKIP_PMC" = "1" ]; then cat $OUT/bench_line.json | cut -c1-300; exit 0; fi
# HBM / fabric bytes per GEMM launch: per shape, isolated (tools/pmc_traffic.sh -> pmc_traffic.json with the private-L2 prediction)
bash tools/pmc_traffic.sh > $OUT/pmc_traffic.log 2>&1
cp gpurun_out/pmc_shapes/pmc_traffic.json $OUT/pmc_traffic.json
# MFMA utilisation / LDS conflicts of the dominant kernel on its two biggest shapes (isolated launches; counters in their own passes)
for shape in fc1 dfc1 wgrp2; do
  tag=$shape
  i=0
  for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_WAIT_INST_ANY" "SQ_WAVES SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY"; do
    rm -rf gpurun_out/prof_gemm_$i
    timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/prof_gemm_$i -- python3 tools/gemm_shape_run.py $shape 10 > /dev/null 2>&1
    i=$((i+1))
  done
  mkdir -p gpurun_out/prof_gemm_all && rm -rf gpurun_out/prof_gemm_all/* && cp -r gpurun_out/prof_gemm_0 gpurun_out/prof_gemm_1 gpurun_out/prof_gemm_2 gpurun_out/prof_gemm_all/
  python3 tools/pmc_summary.py gpurun_out/prof_gemm_all gemm_ > $OUT/gemm_pmc_$tag.json
  rm -rf gpurun_out/prof_gemm_0 gpurun_out/prof_gemm_1 gpurun_out/prof_gemm_2 gpurun_out/prof_gemm_all
done
# the three attention kernels at the step's shape (isolated launches), same three counter sets
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_WAIT_INST_ANY" "SQ_WAVES SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_TRANS_F32"; do
  rm -rf gpurun_out/prof_attn_all/p$i
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/prof_attn_all/p$i -- python3 tools/attn_bench.py > /dev/null 2>&1
  i=$((i+1))
done
python3 tools/pmc_summary.py gpurun_out/prof_attn_all attn_ > $OUT/attn_pmc.json
rm -rf gpurun_out/prof_attn_all
python3 tools/attn_bench.py > $OUT/attn_isolated.txt 2>/dev/null
cat $OUT/bench_line.json | cut -c1-400
# validation decode: bench of the decode forms at three model sizes, isolated kernels, rocprofv3 kernel statistics, fabric bytes
# per launch (separate --pmc passes) and the in-kernel time stamps
if [ "$SKIP_DECODE" != "1" ]; then
  { for m in 410m 160m 1.4b; do timeout -k 10 300 python3 tools/decode_bench.py $m 2>/dev/null; done; echo; timeout -k 10 300 python3 tools/decode_kernel_bench.py 2>/dev/null;
    echo; timeout -k 10 120 python3 tools/decode_ab_trace.py 2>/dev/null; echo; timeout -k 10 120 python3 tools/decode_out_trace.py 2>/dev/null; } > $OUT/decode.txt
  bash tools/decode_profile.sh gpurun_out/decode_prof > /dev/null 2>&1 && cp gpurun_out/decode_prof/decode_kernel_stats.csv $OUT/decode_kernel_stats.csv
  bash tools/decode_pmc.sh gpurun_out/decode_pmc > /dev/null 2>&1 && cp gpurun_out/decode_pmc/decode_pmc.json $OUT/decode_pmc.json
  { timeout -k 10 120 python3 tools/decode_flow_trace.py 2>/dev/null; echo; timeout -k 10 120 python3 tools/decode_pair_trace.py 2>/dev/null; } > $OUT/decode_flow_trace.txt
fi
