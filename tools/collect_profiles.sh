#!/bin/bash
# tools/collect_profiles.sh TAG NAME : copy what tools/profile_round.sh TAG left under gpurun_out/TAG into profiles/NAME_* and regenerate profiles/traffic.json
set -e
TAG=$1; NAME=$2
R=$(cd "$(dirname "$0")/.." && pwd)
G=$R/gpurun_out/$TAG
for c in c2 c1; do
  cp $G/$c/stats_kernel_stats.csv $R/profiles/${NAME}_${c}_kernel_stats.csv
  cp $G/bench_stats_$c.json $R/profiles/${NAME}_bench_${c}_under_rocprof.json
  for p in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU; do cp $G/pmc_${p}_$c.csv $R/profiles/${NAME}_pmc_${p}_$c.csv; done
  python3 $R/profiles/summarize_pmc.py $c $R/profiles/${NAME}_pmc_FETCH_SIZE_$c.csv $R/profiles/${NAME}_pmc_WRITE_SIZE_$c.csv $R/profiles/${NAME}_pmc_SQ_INSTS_VALU_$c.csv
done
for c in c1 c2 c3 c4 c2_exchange_alone c2_3_frames_in_flight; do cp $G/bench_$c.json $R/profiles/${NAME}_bench_$c.json; done
