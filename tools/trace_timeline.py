"""Timeline summary of one training step from a rocprofv3 --kernel-trace CSV: per-queue busy time, the union of all kernel
intervals (how much of the wall clock has at least one kernel running), the depth of overlap, and the per-queue kernel mix.
usage: python tools/trace_timeline.py <kernel_trace.csv> [step_index_from_end=2]"""
import csv
import re
import sys
from collections import defaultdict

path = sys.argv[1]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Queue_Id"]), r["Kernel_Name"]))
rows.sort()
short = lambda n: re.sub(r"\(.*", "", n).replace("void mafed::", "").replace("mafed::", "")[:70]
adam = [i for i, r in enumerate(rows) if "adamw_kernel<" in r[3]]
# a step = (end of the previous step's last adamw launch, end of this step's last adamw launch]
ends = [rows[i][1] for i in adam]
# group adamw launches that are close together (several per step)
marks = [ends[0]]
for e in ends[1:]:
    if e - marks[-1] > 5e6:
        marks.append(e)
    else:
        marks[-1] = e
t0, t1 = marks[-back - 1], marks[-back]
sel = [r for r in rows if r[0] >= t0 and r[1] <= t1 + 1000]
print(f"step window {(t1 - t0) / 1e6:.3f} ms, {len(sel)} kernels")
ev = []
for s, e, q, n in sel:
    ev.append((s, 1))
    ev.append((e, -1))
ev.sort()
depth, last, hist = 0, t0, defaultdict(float)
for t, d in ev:
    hist[depth] += t - last
    last = t
    depth += d
hist[0] += t1 - last
print("overlap depth histogram (ms):", {k: round(v / 1e6, 2) for k, v in sorted(hist.items())})
perq = defaultdict(list)
for r in sel:
    perq[r[2]].append(r)
for q, rs in sorted(perq.items(), key=lambda kv: -sum(r[1] - r[0] for r in kv[1])):
    busy = sum(r[1] - r[0] for r in rs)
    span = rs[-1][1] - rs[0][0]
    print(f"queue {q}: {len(rs)} kernels, busy {busy / 1e6:.2f} ms, first->last {span / 1e6:.2f} ms, starts at +{(rs[0][0] - t0) / 1e6:.2f} ms")
    mix = defaultdict(lambda: [0, 0.0])
    for s, e, _, n in rs:
        m = mix[short(n)]
        m[0] += 1
        m[1] += e - s
    for n, (c, t) in sorted(mix.items(), key=lambda kv: -kv[1][1])[:14]:
        print(f"    {t / 1e6:7.2f} ms {c:5d} x {t / c / 1e3:8.1f} us  {n}")
# idle gaps (no kernel running anywhere) longer than 15 us, with the kernels on either side
iv = sorted((s, e, q, n) for s, e, q, n in sel)
cur_end, cur_name, gaps = iv[0][1], iv[0][3], []
for s, e, q, n in iv[1:]:
    if s > cur_end:
        gaps.append((s - cur_end, cur_end - t0, short(cur_name), short(n), q))
    if e > cur_end:
        cur_end, cur_name = e, n
tot = sum(g[0] for g in gaps)
print(f"idle gaps: {len(gaps)} totalling {tot / 1e6:.2f} ms; > 15 us:")
for g in sorted(gaps, reverse=True)[:25]:
    print(f"   {g[0] / 1e3:8.1f} us at +{g[1] / 1e6:6.2f} ms   after {g[2][:50]}  before {g[3][:50]} (q{g[4]})")
small = [g[0] for g in gaps if g[0] <= 15000]
print(f"   gaps <= 15 us: {len(small)} totalling {sum(small) / 1e6:.2f} ms")
