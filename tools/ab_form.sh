#!/bin/bash
# tools/ab_form.sh V1 V2 .. : integrate kernel time at c2 (culled) and c1 (dense) with RR_K1_FORM=3 and build_variants/lib_V.so ("base" = shipped library, "form2" = shipped library with the default form)
for v in "$@"; do
  unset RGBDR_LIB; export RR_K1_FORM=3
  if [ $v = form2 ]; then unset RR_K1_FORM; elif [ $v != base ]; then export RGBDR_LIB=$PWD/build_variants/lib_$v.so; fi
  python bench.py --no-cpu-baseline --long-steps 0 > gpurun_out/abf_${v}.json 2> gpurun_out/abf_${v}.err
  python -c "import json; d=json.loads(open('gpurun_out/abf_${v}.json').read()); s=d['stage_ms']; c=d['roofline_c1']; print('$v', 'c2 frame', round(d['ms_per_step'],4), 'integrate', round(s['k_integrate_tiles']*1e3,1), '| c1 frame', round(c['ms_per_frame'],4), 'integrate', round(c['integrate']['avg_launch_ms']*1e3,1))"
done
