#!/bin/bash
# timing-only variants of a kernel (build/libplaac_probe<N>.so, results wrong by construction) against the tree's library,
# track mode over the 1.25 M-sequence share, alternating on one box
mkdir -p gpurun_out/r4
out=gpurun_out/r4/ab_probe.txt; : > $out
F="--steps 10 --warmup 3 --no-cpu-baseline --no-e2e --no-predict --no-clock-probe --no-host-leg --tracks --nprot 1250000"
for rep in 1 2; do for v in tree "$@"; do
  if [ $v = tree ]; then unset PLAAC_NATIVE_LIB; else export PLAAC_NATIVE_LIB=$PWD/build/libplaac_$v.so; fi
  timeout -k 10 300 python3 bench.py $F 2>>gpurun_out/r4/ab_probe.err | python3 -c "
import json,sys
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print('$v', 'ms/step', d['ms_per_step'])" >> $out || echo "$v failed" >> $out
done; done
cat $out
