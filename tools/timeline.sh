#!/bin/bash
# tools/timeline.sh OUT SCRIPT [ARGS...] : on the GPU box -- rocprofv3 kernel trace of `python3 SCRIPT ARGS`; prints the kernel timeline of two steady-state frames
set -e
OUT=$1; shift
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$OUT -o tl -- python3 $R/"$@" > $R/gpurun_out/$OUT/stdout.log 2> $R/gpurun_out/$OUT/stderr.log
f=$(find /tmp/tl_$OUT -name '*kernel_trace.csv' | head -n 1)
python3 - "$f" > $R/gpurun_out/$OUT/timeline.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# last-but-30th k_mark_bricks launch starts the window; print 3 frames
marks = [i for i, n in enumerate(names) if "k_mark_bricks" in n]
i0, i1 = marks[-40], marks[-37]
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0:i1]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print(f"{s/1e3:9.1f} {e/1e3:9.1f} {(e-s)/1e3:7.1f}  q{r.get('Queue_Id','?'):>3}  {r['Kernel_Name'][:60]}")
print("frame period us:", (int(rows[marks[-37]]["Start_Timestamp"]) - t0) / 3e3)
PY
cat $R/gpurun_out/$OUT/timeline.txt
