#!/bin/bash
# tools/profile_round.sh TAG : on the GPU box -- kernel-trace stats + FETCH_SIZE / WRITE_SIZE passes (separate runs) of bench.py for c2 and c1
set -e
TAG=$1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for c in c2 c1; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG/$c -o stats -- python3 $R/bench.py --config $c --no-cpu-baseline --steps 100 > $R/gpurun_out/$TAG/bench_stats_$c.json 2> $R/gpurun_out/$TAG/stats_$c.err
  echo "stats $c done"
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/$TAG/$c -o fetch -- python3 $R/bench.py --config $c --no-cpu-baseline --no-timers --steps 20 --warmup 3 > /dev/null 2> $R/gpurun_out/$TAG/fetch_$c.err
  echo "fetch $c done"
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/$TAG/$c -o write -- python3 $R/bench.py --config $c --no-cpu-baseline --no-timers --steps 20 --warmup 3 > /dev/null 2> $R/gpurun_out/$TAG/write_$c.err
  echo "write $c done"
done
cd $R
python3 bench.py > gpurun_out/$TAG/bench_c2.json 2> gpurun_out/$TAG/bench_c2.err
python3 bench.py --config c1 > gpurun_out/$TAG/bench_c1.json 2> gpurun_out/$TAG/bench_c1.err
python3 bench.py --preprocess --no-cpu-baseline > gpurun_out/$TAG/bench_c2_pre.json 2> gpurun_out/$TAG/bench_c2_pre.err
ls -R gpurun_out/$TAG | head -40
