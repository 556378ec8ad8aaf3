#!/bin/bash
# tools/k1_trace.sh [-c CONFIG] V1 V2 .. : on the GPU box -- rocprofv3 kernel-trace average durations (us) of the integrate-side kernels over 300 moving
# frames, per variant library ("default" = the shipped one, NAME = build_variants/lib_NAME.so)
R=$GRAFT_REPO_ROOT
CFG=c2
if [ "$1" = "-c" ]; then CFG=$2; shift 2; fi
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  unset RGBDR_LIB
  [ $v != default ] && export RGBDR_LIB=$R/build_variants/lib_$v.so
  rm -rf /tmp/k1t_$v
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/k1t_$v -o s -- python3 $R/tools/c2_frames.py 300 $CFG > /tmp/k1t_$v.out 2> /tmp/k1t_$v.err || { echo "$v FAILED"; tail -5 /tmp/k1t_$v.err; continue; }
  f=$(find /tmp/k1t_$v -name '*kernel_stats.csv' | head -n 1)
  python3 - "$f" "$v" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
out = []
tot = 0.0
for r in rows:
    n = r["Name"]; calls = int(r["Calls"]); avg = float(r["AverageNs"]) / 1e3
    if calls >= 250:
        tot += avg * calls / 300.0
    if any(k in n for k in ("integrate", "pair_masks", "classify")):
        short = n.split("(")[0].replace("void rr::", "")[:48]
        out.append(f"{short} x{calls} {avg:.1f}")
print(sys.argv[2], "| frame kernels sum %.1f us |" % tot, " | ".join(out), flush=True)
PY
done
