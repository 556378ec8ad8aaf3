#!/bin/bash
# tools/fill_stats.sh TAG [ENV=VAL ...] : on the GPU box -- rocprofv3 kernel stats of the one-stream c2 bench, hole-filling kernels only
TAG=$1; shift
for kv in "$@"; do export "$kv"; done
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp RR_OVERLAP_FILL=0
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG -o stats -- python3 $R/bench.py --config c2 --no-cpu-baseline --no-c1 --long-steps 0 --steps 100 > $R/gpurun_out/$TAG.json 2> $R/gpurun_out/$TAG.err
find $R/gpurun_out/$TAG -name "*kernel_trace.csv" -delete
python3 - <<PY
import csv, glob
f = glob.glob("$R/gpurun_out/$TAG/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if int(r["Calls"]) > 100 and any(k in r["Name"] for k in ("inpaint", "colorfill")): print("$TAG", r["Name"][:44].ljust(46), r["Calls"], "%.2f" % (float(r["AverageNs"]) / 1e3), r["MinNs"], r["MaxNs"])
PY
