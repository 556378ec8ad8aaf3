#!/bin/bash
# tools/pmc_multi.sh KERNEL_SUBSTR "CTR CTR .." ["CTR .." ...] : on the GPU box -- one rocprofv3 --pmc pass per counter group over tools/c2_frames.py
# (60 frames; RGBDR_LIB / RR_* pass through); mean per launch of every counter for kernels whose name contains KERNEL_SUBSTR
KSUB=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for grp in "$@"; do
  rm -rf /tmp/pmc_multi
  rocprofv3 --pmc $grp --output-format csv -d /tmp/pmc_multi -o p -- python3 $R/tools/c2_frames.py 60 ${PMC_CONFIG:-c2} > /dev/null 2> /tmp/pmc_multi.err || { echo "FAILED: $grp"; tail -3 /tmp/pmc_multi.err; continue; }
  f=$(find /tmp/pmc_multi -name '*counter_collection.csv' | head -n 1)
  python3 - "$f" "$KSUB" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r["Kernel_Name"]:
        agg[(r["Kernel_Name"].split("(")[0][-40:], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(agg.items()):
    v = v[len(v) // 2:]
    print(f"{k} {c} mean {sum(v) / len(v):.5g} (n {len(v)})", flush=True)
PY
done
