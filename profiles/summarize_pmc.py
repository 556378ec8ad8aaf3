"""profiles/traffic.json from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (run on the GPU box, separate passes as
MI355X_MICROARCH.md prescribes; both counters are in KiB).

    python profiles/summarize_pmc.py <config> <fetch_counter_collection.csv> <write_counter_collection.csv> [<sq_insts_valu_counter_collection.csv>]

The optional fourth pass (SQ_INSTS_VALU) adds `valu_insts` = VALU wave-instructions per launch; bench.py turns it into
`roofline.valu_issue_frac` = valu_insts x 2 issue cycles (SIMD-32: 32 lanes/cycle, MI355X_MICROARCH.md) / (1024 SIMDs x 2.4 GHz x launch time): the share of the launch the
SIMDs spend issuing vector instructions (the largest single share of the integrate kernel).

Corrections (calibrated on known byte counts with tools/gather_calib.hip, profiles/r02_fetch_calibration_summary.txt):
FETCH_SIZE reports 1/2 of a wide coalesced 16-B-per-lane stream (512 MiB read -> 256 MiB counted), but counts scattered
accesses at their full 64-B sector: 4.00 x the useful bytes for random 16-B taps (128 MiB and 1 GiB tables alike), 1.96 x for
2x2x2 texel neighbourhoods.  The integrate / march / shade kernels are gather kernels (LUT boxes of 3-6 texels per row, scattered
image and voxel taps), so `hbm_bytes` = FETCH_SIZE x 1 + WRITE_SIZE; `hbm_bytes_if_streaming` = FETCH_SIZE x 2 + WRITE_SIZE is
kept as the upper bound that round 1 reported.  WRITE_SIZE is taken as is (calibration on this path: the dense c1 integrate writes 256^3 x 4 B = 67.1 MB and the counter
reads 67.1 MB; a 512 MiB fill reads 524288 KiB)."""
import collections
import csv
import json
import os
import sys

GROUPS = {"k_integrate_tiles_lds": ("k_integrate_tiles_lds", "k_integrate_tiles_rec"),   # (round 4: k_integrate_tiles_rec is the LDS form with the record head; one of the two runs per launch)
          "k_march": ("k_march",), "k_march+k_shade": ("k_march", "k_shade"),                  # k_march covers both passes (k_march<>, k_march_long)
          # round 4: the pre-processing passes (configs named *_preprocess: bench.py --preprocess)
          "k_pre_morph": ("k_pre_morph",), "k_pre_filter": ("k_pre_filter",), "k_pre_boundary": ("k_pre_boundary",), "k_pre_normal": ("k_pre_normal",), "k_pre_quality": ("k_pre_quality",)}


def per_kernel(path, counter=None):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if counter and r["Counter_Name"] != counter:
            continue
        agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v[len(v) // 2:]) / len(v[len(v) // 2:]) for k, v in agg.items()}     # steady-state launches


def per_frame(path):
    """sum over the frame's kernels of (value per launch x launches per frame) from an aggregated pass (tools/pmc_aggregate.py: one row per kernel with its
    launch count); a frame = one k_mark_bricks launch; set-up kernels (fewer launches than frames) are left out"""
    rows = list(csv.DictReader(open(path)))
    if not rows or "Launches" not in rows[0]:
        return None
    frames = max((int(r["Launches"]) for r in rows if "k_mark_bricks" in r["Kernel_Name"]), default=0)
    if not frames:
        return None
    return sum(float(r["Counter_Value"]) * int(r["Launches"]) / frames for r in rows if r["Kernel_Name"].startswith(("rr::", "void rr::")) and int(r["Launches"]) >= frames // 2)


def main():
    cfg, fetch, write = sys.argv[1:4]
    f, w = per_kernel(fetch), per_kernel(write)
    v = per_kernel(sys.argv[4], "SQ_INSTS_VALU") if len(sys.argv) > 4 else {}
    out = {}
    for name, parts in GROUPS.items():
        if not any(any(p in k for p in parts) for k in f):
            continue                                   # the pass did not launch these kernels
        fb = sum(v for k, v in f.items() if any(p in k for p in parts)) * 1024
        wb = sum(v for k, v in w.items() if any(p in k for p in parts)) * 1024
        out[name] = {"fetch_bytes": fb, "write_bytes": wb, "hbm_bytes": fb + wb, "hbm_bytes_if_streaming": 2 * fb + wb}
        if v:
            out[name]["valu_insts"] = sum(x for k, x in v.items() if any(p in k for p in parts))
    ff, wf = per_frame(fetch), per_frame(write)
    if ff is not None and wf is not None:       # the whole frame (bench.py: roofline_frame.traffic); the re-layout kernel streams (FETCH_SIZE counts half of it), the rest gathers
        out["frame"] = {"fetch_bytes": ff * 1024, "write_bytes": wf * 1024, "hbm_bytes": (ff + wf) * 1024, "hbm_bytes_if_streaming": (2 * ff + wf) * 1024}
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "traffic.json")
    data = json.load(open(path)) if os.path.exists(path) else {}
    data[cfg] = out
    json.dump(data, open(path, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
