cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_dec -- python3 tools/decode_bench.py > gpurun_out/dec2.log 2>&1
f=$(find gpurun_out/prof_dec -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY' > gpurun_out/dec_stats.txt
import csv, re, sys
for row in csv.DictReader(open(sys.argv[1])):
    n = re.sub(r"\(.*", "", row["Name"]).replace("void mafed::", "").replace("void at::native::", "at::")[:90]
    print(f'{n:92s} calls {row["Calls"]:>6s} avg_us {float(row["AverageNs"])/1e3:9.1f} total_ms {float(row["TotalDurationNs"])/1e6:9.2f}')
PY
rm -rf gpurun_out/prof_dec
