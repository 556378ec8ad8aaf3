"""FETCH_SIZE (KiB) per launch of the calibration kernels vs their known bytes.  python tools/calib_summary.py <pmc counter_collection.csv>"""
import collections, csv, sys
KNOWN = {"k_calib_stream": 512 * 2 ** 20, "k_calib_gather23": 128 * 2 ** 20, "k_calib_gather26": 2 ** 30, "k_calib_tap8": (1 << 21) * 8 * 16}
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] == "FETCH_SIZE":
        agg[r["Kernel_Name"].split("(")[0].split("<")[0].split("::")[-1]].append(float(r["Counter_Value"]))
for k, known in KNOWN.items():
    v = agg.get(k)
    if not v:
        continue
    kib = sum(v) / len(v)
    print(f"{k:18s} launches {len(v):3d}  FETCH_SIZE {kib * 1024 / 2**20:10.1f} MiB   known useful bytes {known / 2**20:8.1f} MiB   FETCH_SIZE / useful = {kib * 1024 / known:.3f}")
