#!/bin/bash
# round 6: summary mode (the 10 M-sequence headline step), same box: an older build (OLD_LIB) against the tree's, alternating
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r6; mkdir -p $O; TAG=${TAG:-sum}
out=$O/summary_${TAG}.txt; : > $out
F="--steps 20 --warmup 3 --no-cpu-baseline --no-e2e --no-predict --no-clock-probe --no-host-leg --no-tracks-leg --no-tolerance-leg"
run() { L=$1; shift
  env "$@" timeout -k 10 300 python3 bench.py $F 2>>$O/summary_${TAG}.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L:', 'ms/step', d['ms_per_step'], {k: round(x,3) for k,x in d['roofline']['kernel_ms'].items()})" >> $out || echo "$L failed" >> $out
}
OLD=$GRAFT_REPO_ROOT/${OLD_LIB:-build/libplaac_r05.so}
for rep in 1 2 3; do
  run "old        " PLAAC_NATIVE_LIB=$OLD
  run "new        " X=1
done
run "old serial " PLAAC_NATIVE_LIB=$OLD PLAAC_SERIAL_STREAMS=1
run "new serial " PLAAC_SERIAL_STREAMS=1
cat $out
