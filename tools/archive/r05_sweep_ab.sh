#!/bin/bash
# round 5: the 9-point sweep over the 1.25 M share, same box: core search of the long wave-groups aside, chains enqueued first
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r5; mkdir -p $O
out=$O/sweep_ab.txt; : > $out
F="--steps 12 --warmup 3 --no-cpu-baseline --no-e2e --no-predict --no-clock-probe --no-host-leg --no-tracks-leg --sweep --nprot 1250000"
run() { L=$1; shift
  env "$@" timeout -k 10 300 python3 bench.py $F 2>>$O/sweep_ab.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L:', 'ms/step', d['ms_per_step'])" >> $out || echo "$L failed" >> $out
}
for rep in 1 2 3; do
  run "rounds 3 - 4               " PLAAC_SWEEP_CORE_ASIDE=0 PLAAC_SWEEP_REST_ASIDE=0
  run "core aside, one vit stream " PLAAC_SWEEP_REST_ASIDE=0
  run "core + rest aside          " PLAAC_SWEEP_REST_ASIDE=1
  run "the same, chains first     " PLAAC_SWEEP_REST_ASIDE=1 PLAAC_SWEEP_CHAINS_FIRST=1
done
cat $out
