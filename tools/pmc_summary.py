"""Sum rocprofv3 --pmc counter_collection CSVs per kernel name: python tools/pmc_summary.py <dir> [name filter] -> JSON on stdout."""
import csv
import glob
import json
import re
import sys
from collections import defaultdict

root = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    with open(f) as fp:
        for row in csv.DictReader(fp):
            name = re.sub(r"\(.*", "", row["Kernel_Name"]).replace("void mafed::", "")
            if flt and flt not in name:
                continue
            a = acc[name][row["Counter_Name"]]
            a[0] += float(row["Counter_Value"])
            a[1] += 1
out = {k: {c: round(v[0] / max(v[1], 1), 1) for c, v in sorted(cs.items())} for k, cs in acc.items()}
for k, c in out.items():   # per-wave instruction mix and pipe occupancy, where the counters for them were collected
    d, w = {}, c.get("SQ_WAVES", 0.0)
    if w:
        for name, key in (("mfma", "SQ_INSTS_MFMA"), ("valu", "SQ_INSTS_VALU"), ("lds", "SQ_INSTS_LDS"), ("trans", "SQ_INSTS_VALU_TRANS_F32")):
            if key in c:
                d[name + "_insts_per_wave"] = round(c[key] / w)
    wc = c.get("SQ_WAVE_CYCLES", 0.0)
    if wc:
        for name, key in (("valu_active", "SQ_ACTIVE_INST_VALU"), ("wait_inst", "SQ_WAIT_INST_ANY"), ("wait_any", "SQ_WAIT_ANY")):
            if key in c:
                d[name + "_frac_of_wave_cycles"] = round(c[key] / wc, 3)
    if c.get("SQ_BUSY_CYCLES") and "SQ_VALU_MFMA_BUSY_CYCLES" in c:
        # SQ_BUSY_CYCLES sums the 32 shader engines, SQ_VALU_MFMA_BUSY_CYCLES the 1024 SIMDs
        d["mfma_busy_frac_of_simd_cycles"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["SQ_BUSY_CYCLES"] / 32 * 1024), 3)
    if c.get("SQ_ACTIVE_INST_LDS") and "SQ_LDS_BANK_CONFLICT" in c:
        d["lds_bank_conflict_cycles_per_lds_active_cycle"] = round(c["SQ_LDS_BANK_CONFLICT"] / c["SQ_ACTIVE_INST_LDS"], 3)
    if d:
        c["derived"] = d
print(json.dumps(out, indent=1))
