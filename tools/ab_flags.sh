#!/bin/bash
# tools/ab_flags.sh V1 V2 .. : bench c2 and c1 with build_variants/lib_V.so (and "base" = the shipped library)
for v in base "$@"; do
  if [ $v = base ]; then unset RGBDR_LIB; else export RGBDR_LIB=$PWD/build_variants/lib_$v.so; fi
  for c in c2 c1; do
    python bench.py --no-cpu-baseline --config $c > gpurun_out/ab_${v}_$c.json 2> gpurun_out/ab_${v}_$c.err
    python -c "import json; d=json.loads(open('gpurun_out/ab_${v}_$c.json').read()); s=d['stage_ms']; print('$v $c', round(d['ms_per_step'],4), {k: round(x,4) for k,x in s.items()})"
  done
done
