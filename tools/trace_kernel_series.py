"""Durations of chosen kernels launch by launch from a rocprofv3 kernel trace (CSV), oldest first:
python tools/trace_kernel_series.py <kernel_trace.csv> <last> <name> [<name> ...]  -- prints the
last `last` launches of every kernel whose name contains <name>, one row per launch index."""
import csv
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
last = int(sys.argv[2])
names = sys.argv[3:]
series = []
for nm in names:
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if nm in r["Kernel_Name"]]
    series.append(d[-last:])
print("launch  " + "  ".join("%14s" % nm[:14] for nm in names))
for i in range(min(len(s) for s in series)):
    print("%6d  " % (i - last) + "  ".join("%14.1f" % s[i] for s in series))
