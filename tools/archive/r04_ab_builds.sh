#!/bin/bash
# tools/r04_ab_builds.sh [bench args] - same-box A/B: build/libplaac_old.so against the tree's library, alternating, three rounds
mkdir -p gpurun_out/r4
out=gpurun_out/r4/ab_builds.txt; : > $out
F="--steps 16 --warmup 4 --no-cpu-baseline --no-e2e --no-predict --no-clock-probe --no-host-leg"
for rep in 1 2 3; do for v in old new; do
  if [ $v = old ]; then export PLAAC_NATIVE_LIB=$PWD/build/libplaac_old.so; else unset PLAAC_NATIVE_LIB; fi
  timeout -k 10 300 python3 bench.py $F "$@" 2>>gpurun_out/r4/ab_builds.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', 'ms/step', d['ms_per_step'])" >> $out || echo "$v failed" >> $out
done; done
cat $out
