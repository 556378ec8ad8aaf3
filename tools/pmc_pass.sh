#!/bin/bash
# tools/pmc_pass.sh TAG CONFIG "COUNTER COUNTER ..." : one rocprofv3 --pmc pass of bench.py (GPU box), csv under gpurun_out/TAG/
set -e
TAG=$1; CFG=$2; CTRS=$3; NAME=$4
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $CTRS --output-format csv -d $R/gpurun_out/$TAG -o $NAME -- python3 $R/bench.py --config $CFG --no-cpu-baseline --no-timers --steps 10 --warmup 3 > /dev/null 2> $R/gpurun_out/$TAG/$NAME.err
echo "$NAME done"
