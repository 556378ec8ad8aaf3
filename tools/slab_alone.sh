#!/bin/bash
# tools/slab_alone.sh N [config] (EVEN_ONLY=1: equal thickness only) : what each worker slab of an N-slab partition costs per frame on ONE GPU (bench.py rehearsal hooks), equal-thickness
# and balanced boundaries: volume-side stage times (bricks + integrate + depth limits + draw), i.e. without the compositing rank's share
N=$1; CFG=${2:-c2}
for mode in "" "/balanced"; do
  if [ -n "$EVEN_ONLY" ] && [ -n "$mode" ]; then continue; fi
  for k in $(seq 0 $((N - 1))); do
    RR_BENCH_EXCHANGE_ALONE=1 RR_BENCH_ALONE_SLAB=$k/$N$mode python bench.py --config $CFG --no-cpu-baseline --no-c1 --long-steps 0 --steps 100 > gpurun_out/slab_alone.json 2> gpurun_out/slab_alone.err || { tail -5 gpurun_out/slab_alone.err; exit 1; }
    python -c "
import json; d=json.load(open('gpurun_out/slab_alone.json')); s=d['stage_ms']; v=s['bricks']+s['2integrate']+s.get('brickdraw',0)+s['draw']
ahead = s.get('0repack', 0) + s['bricks']          # round 3: on the lane ahead, beside the previous frame's kernels
print('$k/$N$mode', d['slab_check'].split('planes ')[1].split(')')[0], 'volume side (one stream) %.1f us, of it on the lane ahead %.1f' % (v*1e3 + s.get('0repack', 0)*1e3, ahead*1e3), {k: round(x*1e3,1) for k,x in s.items() if k in ('0repack','bricks','2integrate','k_integrate_tiles','brickdraw','draw','k_march')}, 'frame: one stream %.1f, three lanes %.1f us (with this rank compositing as well)' % (d['serial']['ms_per_step']*1e3, d['ms_per_step']*1e3))"
  done
done
