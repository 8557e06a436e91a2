cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4r
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r4r/dec -- python3 $GRAFT_REPO_ROOT/tools/decode_bench.py > $GRAFT_REPO_ROOT/gpurun_out/r4r/decode_prof.log 2>&1
cd $GRAFT_REPO_ROOT
ks=$(find gpurun_out/r4r/dec -name "*kernel_stats.csv" | head -1)
head -40 $ks | cut -c1-260 > gpurun_out/r4r/decode_kernel_stats.txt
kt=$(find gpurun_out/r4r/dec -name "*kernel_trace.csv" | head -1)
python3 - $kt <<'PY' > gpurun_out/r4r/decode_tail_trace.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
tail = rows[-160:]
t0 = int(tail[0]["Start_Timestamp"])
for r in tail:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print(f"{s/1e3:9.2f} {(e-s)/1e3:7.2f} {r['Kernel_Name'][:110]} grid={r.get('Grid_Size_X','?')} wg={r.get('Workgroup_Size_X','?')}")
PY
rm -rf gpurun_out/r4r/dec
cat gpurun_out/r4r/decode_kernel_stats.txt | cut -c1-200
