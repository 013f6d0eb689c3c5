"""Per-kernel time per step from a rocprofv3 kernel trace (CSV) of a run whose last launches are whole
Barnes-Hut steps: python tools/trace_steps.py <kernel_trace.csv> [steps]  -- averages over the last
`steps` steps (a step ends with the traversal kernel)."""
import collections
import csv
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ends = [i for i, r in enumerate(rows) if "_force<" in r["Kernel_Name"] and "false>" in r["Kernel_Name"]]
a, b = ends[-steps - 1] + 1, ends[-1] + 1
agg = collections.OrderedDict()
for r in rows[a:b]:
    n = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("nb::", "")
    agg.setdefault(n, [0, 0])
    agg[n][0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    agg[n][1] += 1
wall = (int(rows[b - 1]["End_Timestamp"]) - int(rows[a]["Start_Timestamp"])) / steps / 1e3
busy = sum(t for t, _ in agg.values()) / steps / 1e3
print("last %d steps: %.1f us per step wall, %.1f us in kernels, %d launches per step" % (steps, wall, busy, (b - a) // steps))
for n, (t, c) in agg.items():
    print("   %-44s %9.1f us/step  (%.1f launches/step)" % (n[:44], t / steps / 1e3, c / steps))
