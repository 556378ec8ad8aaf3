"""Shrinks a rocprofv3 --pmc counter_collection.csv (one row per launch: megabytes for a bench run) to one row per
(kernel, counter): the mean over the steady-state half of the launches -- the same reduction profiles/summarize_pmc.py applies --
plus the launch count.  Column names stay those of rocprofv3 so that the other scripts read either file.

    python tools/pmc_aggregate.py <counter_collection.csv> <out.csv>"""
import collections
import csv
import sys

agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    agg[(r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
with open(sys.argv[2], "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Kernel_Name", "Counter_Name", "Counter_Value", "Launches", "Min", "Max"])
    for (k, c), v in sorted(agg.items()):
        s = v[len(v) // 2:]
        w.writerow([k, c, repr(sum(s) / len(s)), len(v), repr(min(s)), repr(max(s))])
