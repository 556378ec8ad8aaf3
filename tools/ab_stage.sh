#!/bin/bash
# tools/ab_stage.sh V1 V2 .. : c2 bench stage times with build_variants/lib_V.so ("base" = the shipped library), three runs each
for rep in 1 2 3; do
for v in "$@"; do
  if [ $v = base ]; then unset RGBDR_LIB; else export RGBDR_LIB=$PWD/build_variants/lib_$v.so; fi
  python bench.py --no-cpu-baseline --no-c1 --long-steps 0 > gpurun_out/ab_${v}.json 2> gpurun_out/ab_${v}.err
  python -c "import json; d=json.loads(open('gpurun_out/ab_${v}.json').read()); s=d['stage_ms']; print('$v', round(d['ms_per_step'],4), round(d['static']['ms_per_step'],4), {k: round(x*1e3,1) for k,x in s.items()})"
done
done
