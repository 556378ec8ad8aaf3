#!/bin/bash
# tools/ab_cap.sh LIBVARIANT CAP... : c2 bench with RR_MARCH_CAP sweeps for one build variant
v=$1; shift
if [ $v = base ]; then unset RGBDR_LIB; else export RGBDR_LIB=$PWD/build_variants/lib_$v.so; fi
for cap in "$@"; do
  RR_MARCH_CAP=$cap python bench.py --no-cpu-baseline > gpurun_out/abc_${v}_$cap.json 2> /dev/null
  python -c "import json; d=json.loads(open('gpurun_out/abc_${v}_$cap.json').read()); s=d['stage_ms']; print('$v cap $cap', round(d['ms_per_step'],4), 'march', round(s['k_march'],4), 'draw', round(s['draw'],4))"
done
