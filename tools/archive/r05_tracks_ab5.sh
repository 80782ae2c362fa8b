#!/bin/bash
# round 5: track mode, same box: forms of the forward pass with the posteriors
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r5; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "track_mode_posteriors" > $O/ab5_tests.txt 2>&1; tail -n 5 $O/ab5_tests.txt
grep -q "passed" $O/ab5_tests.txt && ! grep -q "failed" $O/ab5_tests.txt || exit 1
out=$O/tracks_ab5.txt; : > $out
F="--steps 16 --warmup 4 --no-cpu-baseline --no-e2e --no-predict --no-clock-probe --no-host-leg --no-tracks-leg --tracks --nprot 1250000"
run() { L=$1; shift
  env "$@" timeout -k 10 300 python3 bench.py $F 2>>$O/tracks_ab5.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L:', 'ms/step', d['ms_per_step'], d['roofline']['kernel_ms'])" >> $out || echo "$L failed" >> $out
}
for rep in 1 2 3; do
  run "lane stores, occ 3 " PLAAC_TRACK_POST_FORM=0 PLAAC_TRACK_KB_LATE=0
  run "transposed 256     " PLAAC_TRACK_POST_FORM=1 PLAAC_TRACK_KB_LATE=0
  run "transposed 768     " PLAAC_TRACK_POST_FORM=2 PLAAC_TRACK_KB_LATE=0
done
cat $out
