for v in base s3 s4 m3 m4; do
  if [ $v = base ]; then unset RGBDR_LIB; else export RGBDR_LIB=$PWD/build_variants/lib_$v.so; fi
  python bench.py --no-cpu-baseline > gpurun_out/ab_$v.json 2> gpurun_out/ab_$v.err
  python -c "import json; d=json.loads(open('gpurun_out/ab_$v.json').read()); s=d['stage_ms']; print('$v', round(d['ms_per_step'],4), 'march', round(s.get('k_march',0),4), 'draw', round(s.get('draw',0),4))"
done
