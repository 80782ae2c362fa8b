#!/bin/bash
# tools/r04_ab_env.sh "<label A>" "<env A>" "<label B>" "<env B>" [bench args] - same-box A/B of two environment settings, alternating, three rounds
mkdir -p gpurun_out/r4
out=gpurun_out/r4/ab_env.txt; : > $out
LA=$1; EA=$2; LB=$3; EB=$4; shift 4
F="--steps 16 --warmup 4 --no-cpu-baseline --no-e2e --no-predict --no-clock-probe --no-host-leg"
for rep in 1 2 3; do for v in A B; do
  if [ $v = A ]; then L=$LA; E=$EA; else L=$LB; E=$EB; fi
  env $E timeout -k 10 300 python3 bench.py $F "$@" 2>>gpurun_out/r4/ab_env.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L:', 'ms/step', d['ms_per_step'])" >> $out || echo "$L failed" >> $out
done; done
cat $out
