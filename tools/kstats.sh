#!/bin/bash
# tools/kstats.sh OUT SCRIPT [ARGS...] : on the GPU box -- rocprofv3 kernel-trace stats of `python3 SCRIPT ARGS`, top kernels printed and kept under gpurun_out/OUT
set -e
OUT=$1; shift
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$OUT -o stats -- python3 $R/"$@" > $R/gpurun_out/$OUT/stdout.log 2> $R/gpurun_out/$OUT/stderr.log
find $R/gpurun_out/$OUT -name '*kernel_trace.csv' -delete
f=$(find $R/gpurun_out/$OUT -name '*kernel_stats.csv' | head -n 1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.reader(open(sys.argv[1])))
for r in rows[:12]:
    print(r[0][:70], r[1:5])
PY
