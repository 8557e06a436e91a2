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
print(json.dumps(out, indent=1))
