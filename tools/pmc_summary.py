"""Per-kernel averages of a rocprofv3 --pmc counter_collection.csv (steady-state half of the launches)."""
import collections, csv, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(agg.items()):
    if not any(s in k for s in ("k_integrate_tiles", "k_march", "k_shade", "k_inpaint", "k_depth")):
        continue
    print(k)
    for c, v in sorted(d.items()):
        v = v[len(v) // 2:]
        print(f"    {c:34s} {sum(v) / len(v):16.1f}")
