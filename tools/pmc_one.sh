#!/bin/bash
# tools/pmc_one.sh COUNTER KERNEL_SUBSTR SCRIPT [ARGS...] : rocprofv3 --pmc COUNTER of `python3 SCRIPT ARGS` (environment passes through); mean / min / max of the
# counter over the launches of kernels whose name contains KERNEL_SUBSTR
set -e
CTR=$1; KSUB=$2; shift 2
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_one
rocprofv3 --pmc $CTR --output-format csv -d /tmp/pmc_one -o p -- python3 $R/"$@" > /dev/null 2> /tmp/pmc_one.err || { tail -5 /tmp/pmc_one.err; exit 1; }
f=$(find /tmp/pmc_one -name '*counter_collection.csv' | head -n 1)
python3 - "$f" "$CTR" "$KSUB" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] == sys.argv[2] and sys.argv[3] in r["Kernel_Name"]:
        agg[r["Kernel_Name"][:60]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    v = v[len(v) // 2:]
    print(sys.argv[2], k, "launches", len(v), "mean %.4g min %.4g max %.4g" % (sum(v) / len(v), min(v), max(v)))
PY
