#!/bin/bash
# tools/profile_round.sh TAG : on the GPU box -- kernel-trace stats + FETCH_SIZE / WRITE_SIZE / SQ_INSTS_VALU passes (separate runs) of
# bench.py for c2 and c1, then the plain bench lines.  The --pmc passes are reduced to one row per (kernel, counter) by
# tools/pmc_aggregate.py (the raw CSVs hold one row per launch, megabytes per pass).
# Round 3: the kernel-trace stats are taken twice for c2 -- as shipped (all lanes: kernels stretched by their co-runners) and with
# RR_OVERLAP_FILL=0 (one stream: the kernel times bench.py's roofline uses); the counter passes run on one stream.
set -e
TAG=$1
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
# round 4: the counter passes also for configs[3] / configs[4] (their roofline.traffic was null), one pass of the texture-address /
# vector-memory counters for c2 and c4, and the kernel stats of the pre-processing passes (bench.py --preprocess)
for c in c2 c1 c3 c4; do
  for mode in lanes serial; do
    [ $mode = serial ] && export RR_OVERLAP_FILL=0 || unset RR_OVERLAP_FILL
    [ $c != c2 ] && [ $mode = lanes ] && continue
    rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG/${c}_$mode -o stats -- python3 $R/bench.py --config $c --no-cpu-baseline --no-c1 --long-steps 0 --steps 100 > $R/gpurun_out/$TAG/bench_stats_${c}_$mode.json 2> $R/gpurun_out/$TAG/stats_${c}_$mode.err
    find $R/gpurun_out/$TAG/${c}_$mode -name '*kernel_trace.csv' -delete
    echo "stats $c $mode done"
  done
  export RR_OVERLAP_FILL=0
  for pmc in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU; do
    rocprofv3 --pmc $pmc --output-format csv -d /tmp/pmc_$c -o $pmc -- python3 $R/bench.py --config $c --no-cpu-baseline --no-timers --no-c1 --long-steps 0 --steps 20 --warmup 3 > /dev/null 2> $R/gpurun_out/$TAG/${pmc}_$c.err
    f=$(find /tmp/pmc_$c -name "${pmc}_counter_collection.csv" | head -n 1)
    python3 $R/tools/pmc_aggregate.py $f $R/gpurun_out/$TAG/pmc_${pmc}_$c.csv
    echo "$pmc $c done"
  done
  if [ $c = c2 ] || [ $c = c4 ]; then
    for pmc in TA_BUSY_avr SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_WAVE_CYCLES; do
      rocprofv3 --pmc $pmc --output-format csv -d /tmp/pmc2_$c -o $pmc -- python3 $R/bench.py --config $c --no-cpu-baseline --no-timers --no-c1 --long-steps 0 --steps 20 --warmup 3 > /dev/null 2> $R/gpurun_out/$TAG/${pmc}_$c.err || true
      f=$(find /tmp/pmc2_$c -name "${pmc}_counter_collection.csv" | head -n 1)
      [ -n "$f" ] && python3 $R/tools/pmc_aggregate.py $f $R/gpurun_out/$TAG/pmc_${pmc}_$c.csv || true
      echo "$pmc $c done"
    done
  fi
  unset RR_OVERLAP_FILL
done
RR_OVERLAP_FILL=0 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG/c2_preprocess -o stats -- python3 $R/bench.py --preprocess --no-cpu-baseline --no-c1 --long-steps 0 --steps 100 > $R/gpurun_out/$TAG/bench_stats_c2_preprocess.json 2> $R/gpurun_out/$TAG/stats_c2_preprocess.err
find $R/gpurun_out/$TAG/c2_preprocess -name '*kernel_trace.csv' -delete
export RR_OVERLAP_FILL=0
for pmc in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU; do
  rocprofv3 --pmc $pmc --output-format csv -d /tmp/pmc_pre -o $pmc -- python3 $R/bench.py --preprocess --no-cpu-baseline --no-timers --no-c1 --long-steps 0 --steps 20 --warmup 3 > /dev/null 2> $R/gpurun_out/$TAG/${pmc}_c2_preprocess.err
  f=$(find /tmp/pmc_pre -name "${pmc}_counter_collection.csv" | head -n 1)
  python3 $R/tools/pmc_aggregate.py $f $R/gpurun_out/$TAG/pmc_${pmc}_c2_preprocess.csv
done
unset RR_OVERLAP_FILL
echo "stats preprocess done"
cd $R
python3 bench.py > gpurun_out/$TAG/bench_c2.json 2> gpurun_out/$TAG/bench_c2.err
echo "bench c2 done"
python3 bench.py --config c1 --no-cpu-baseline > gpurun_out/$TAG/bench_c1.json 2> gpurun_out/$TAG/bench_c1.err
python3 bench.py --config c3 --no-cpu-baseline --no-c1 > gpurun_out/$TAG/bench_c3.json 2> gpurun_out/$TAG/bench_c3.err
python3 bench.py --config c4 --no-cpu-baseline --no-c1 > gpurun_out/$TAG/bench_c4.json 2> gpurun_out/$TAG/bench_c4.err
echo "bench c1 c3 c4 done"
RR_BENCH_EXCHANGE_ALONE=1 python3 bench.py --no-cpu-baseline --no-c1 --long-steps 0 > gpurun_out/$TAG/bench_c2_exchange_alone.json 2> gpurun_out/$TAG/bench_c2_exchange_alone.err
RR_BENCH_EXCHANGE_ALONE=1 python3 bench.py --no-cpu-baseline --no-c1 --long-steps 0 --exchange native > gpurun_out/$TAG/bench_c2_exchange_alone_native.json 2> gpurun_out/$TAG/bench_c2_exchange_alone_native.err
python3 bench.py --preprocess --no-cpu-baseline --no-c1 --long-steps 0 > gpurun_out/$TAG/bench_c2_preprocess.json 2> gpurun_out/$TAG/bench_c2_preprocess.err
RR_OVERLAP_FILL=0 python3 bench.py --no-cpu-baseline --no-c1 --long-steps 0 > gpurun_out/$TAG/bench_c2_one_stream.json 2> gpurun_out/$TAG/bench_c2_one_stream.err
python3 bench.py --frames-in-flight 3 --no-cpu-baseline --no-c1 --long-steps 0 > gpurun_out/$TAG/bench_c2_3_frames_in_flight.json 2> gpurun_out/$TAG/bench_c2_3fif.err
echo "all done"
# round 4: the wire path (f2) -- host message -> pinned copy -> H2D -> GPU unpack / DXT decode -> pre-processing -> the frame
for f in f32-rgb8 u8-dxt1; do
  python3 bench.py --ingest $f --no-cpu-baseline --no-c1 --long-steps 0 > gpurun_out/$TAG/bench_c2_ingest_$f.json 2> gpurun_out/$TAG/bench_c2_ingest_$f.err || true
done
echo "ingest done"
