#!/bin/bash
# timing-only variants (build/libplaac_<name>.so, results wrong by construction) against the tree's library on the
# headline workload (10 M sequences, overlapping steps), alternating on one box
mkdir -p gpurun_out/r4
out=gpurun_out/r4/ab_probe_full.txt; : > $out
F="--steps 16 --warmup 4 --no-cpu-baseline --no-e2e --no-predict --no-clock-probe --no-host-leg"
for rep in 1 2 3; do for v in tree "$@"; do
  if [ $v = tree ]; then unset PLAAC_NATIVE_LIB; else export PLAAC_NATIVE_LIB=$PWD/build/libplaac_$v.so; fi
  timeout -k 10 300 python3 bench.py $F 2>>gpurun_out/r4/ab_probe_full.err | python3 -c "
import json,sys
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print('$v', 'ms/step', d['ms_per_step'])" >> $out || echo "$v failed" >> $out
done; done
cat $out
