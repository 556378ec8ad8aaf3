#!/bin/bash
# tools/rehearse_n.sh N REPS [bench args] : bench.py --gpus N on ONE GPU (gloo standing in for RCCL), REPS times; prints rc and the error tail of failures
N=$1; REPS=$2; shift 2
for r in $(seq 1 $REPS); do
  RR_BENCH_BACKEND=gloo RR_BENCH_DEVICE=0 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29700 + r)) bench.py --gpus $N --steps 10 --warmup 2 --long-steps 0 --no-cpu-baseline "$@" > gpurun_out/reh_$r.json 2> gpurun_out/reh_$r.err
  rc=$?
  echo "run $r rc=$rc $(python -c "import json;d=json.load(open('gpurun_out/reh_$r.json'));print(round(d['value'],1), d['regathers'])" 2>/dev/null)"
  if [ $rc != 0 ]; then grep -v "amdgpu.ids\|socket.cpp\|Gloo\|^\*\|OMP_NUM" gpurun_out/reh_$r.err | tail -15; fi
done
