"""Idle time between consecutive kernels of one stream, from a rocprofv3 --kernel-trace CSV:
python profiles/gaps.py <kernel_trace.csv> [first_kernel_name_substring]
Prints, for the steps delimited by the named kernel (default k_bbox_partial), the mean wall time of a
step, the mean sum of its kernels' durations and the mean idle time between them."""
import csv
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
first = sys.argv[2] if len(sys.argv) > 2 else "k_bbox_partial"
starts = [i for i, r in enumerate(rows) if first in r[2]]
steps = []
for a, b in zip(starts[:-1], starts[1:]):
    seg = rows[a:b]
    wall = rows[b][0] - seg[0][0]
    busy = sum(e - s for s, e, _ in seg)
    steps.append((wall, busy, len(seg)))
# the steady steps: the most common kernel count
from collections import Counter
common = Counter(s[2] for s in steps).most_common(1)[0][0]
sel = [s for s in steps if s[2] == common]
sel = sel[len(sel) // 4:]
n = len(sel)
if "--last" in sys.argv:
    # per-kernel mean time over the last steps of the trace (whatever their kernel count)
    k = int(sys.argv[sys.argv.index("--last") + 1])
    per = {}
    a0 = starts[-k - 1]
    for s0, e0, name in rows[a0:starts[-1]]:
        short = name.split("(")[0].replace("void ", "").replace("nb::", "")
        c = per.setdefault(short, [0, 0])
        c[0] += 1
        c[1] += e0 - s0
    for name, (cnt, tot) in sorted(per.items(), key=lambda kv: -kv[1][1]):
        print("  %-34s %5.1f launches/step  %8.1f us/step" % (name[:34], cnt / k, tot / k / 1e3))
print("steps with %d kernels: %d used; wall %.1f us, kernels %.1f us, idle between kernels %.1f us (%.1f us per launch)" % (
    common, n, sum(s[0] for s in sel) / n / 1e3, sum(s[1] for s in sel) / n / 1e3,
    sum(s[0] - s[1] for s in sel) / n / 1e3, sum(s[0] - s[1] for s in sel) / n / 1e3 / common))
