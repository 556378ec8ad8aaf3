#!/bin/bash
# tools/collect_profiles.sh TAG NAME : copy what tools/profile_round.sh TAG left under gpurun_out/TAG into profiles/NAME_* and regenerate profiles/traffic.json
set -e
TAG=$1; NAME=$2
R=$(cd "$(dirname "$0")/.." && pwd)
G=$R/gpurun_out/$TAG
find_stats() { find $G/$1 -name '*kernel_stats.csv' | head -n 1; }
# the kernel stats the roofline figures agree with: one stream (RR_OVERLAP_FILL=0); the run with all lanes (as shipped) beside them
cp $(find_stats c2_serial) $R/profiles/${NAME}_c2_kernel_stats.csv
cp $(find_stats c2_lanes) $R/profiles/${NAME}_c2_all_lanes_kernel_stats.csv
cp $(find_stats c1_serial) $R/profiles/${NAME}_c1_kernel_stats.csv
cp $G/bench_stats_c2_serial.json $R/profiles/${NAME}_bench_c2_under_rocprof.json
cp $G/bench_stats_c2_lanes.json $R/profiles/${NAME}_bench_c2_all_lanes_under_rocprof.json
cp $G/bench_stats_c1_serial.json $R/profiles/${NAME}_bench_c1_under_rocprof.json
cp $(find_stats c3_serial) $R/profiles/${NAME}_c3_kernel_stats.csv
cp $(find_stats c4_serial) $R/profiles/${NAME}_c4_kernel_stats.csv
cp $(find_stats c2_preprocess) $R/profiles/${NAME}_c2_preprocess_kernel_stats.csv
for c in c2 c4; do for p in TA_BUSY_avr SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_WAVE_CYCLES; do [ -f $G/pmc_${p}_$c.csv ] && cp $G/pmc_${p}_$c.csv $R/profiles/${NAME}_pmc_${p}_$c.csv || true; done; done
for c in c2 c1 c3 c4 c2_preprocess; do
  for p in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU; do cp $G/pmc_${p}_$c.csv $R/profiles/${NAME}_pmc_${p}_$c.csv; done
  python3 $R/profiles/summarize_pmc.py $c $R/profiles/${NAME}_pmc_FETCH_SIZE_$c.csv $R/profiles/${NAME}_pmc_WRITE_SIZE_$c.csv $R/profiles/${NAME}_pmc_SQ_INSTS_VALU_$c.csv
done
for c in c1 c2 c3 c4 c2_preprocess c2_exchange_alone c2_exchange_alone_native c2_one_stream c2_3_frames_in_flight; do cp $G/bench_$c.json $R/profiles/${NAME}_bench_$c.json; done
for f in f32-rgb8 u8-dxt1; do [ -f $G/bench_c2_ingest_$f.json ] && cp $G/bench_c2_ingest_$f.json $R/profiles/${NAME}_bench_c2_ingest_$f.json || true; done
