cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_tr
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_tr -- python3 bench.py --no-cpu-baseline --no-kernel-profile --no-secondary --no-image-leg --no-teacher-cache-leg --steps 6 --warmup 4 > /dev/null 2>&1
kt=$(find gpurun_out/prof_tr -name "*kernel_trace.csv" | head -1)
python3 - "$kt" <<'PY'
import csv, re, sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
short=lambda n: re.sub(r"\(.*","",n).replace("void mafed::","").replace("void at::native::","at::")[:70]
idx=[i for i,r in enumerate(rows) if "ce_fwd_kernel" in r["Kernel_Name"]]
i=idx[-2]
t0=int(rows[i]["Start_Timestamp"])
for r in rows[i-6:i+60]:
    s=int(r["Start_Timestamp"]); e=int(r["End_Timestamp"])
    print(f'{(s-t0)/1e3:9.1f} us  +{(e-s)/1e3:7.1f}  q{r["Queue_Id"]}  {short(r["Kernel_Name"])}')
PY
rm -rf gpurun_out/prof_tr
