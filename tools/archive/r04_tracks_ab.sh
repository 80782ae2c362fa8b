#!/bin/bash
mkdir -p gpurun_out/r4
out=gpurun_out/r4/tracks_ab.txt
: > $out
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "track or forms or runs or overlap"  > gpurun_out/r4/tracks_pytest.txt 2>&1 || { tail -30 gpurun_out/r4/tracks_pytest.txt; exit 1; }
tail -2 gpurun_out/r4/tracks_pytest.txt
F="--steps 10 --warmup 3 --no-cpu-baseline --no-e2e --no-predict --no-clock-probe --no-host-leg --tracks --nprot 1250000"
run() { echo "== $1" >> $out; shift
  env "$@" timeout -k 10 300 python3 bench.py $F 2>>gpurun_out/r4/tracks_ab.err | python3 -c "
import sys,json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print('   ms_per_step', d['ms_per_step'])
" >> $out || echo "   failed" >> $out; }
OLD="PLAAC_TRACK_FUSED=0 PLAAC_TRACK_VIT_MIXED=0 PLAAC_TRACK_SEGMENTS=2"
run "round 3's forms" $OLD
run "default" PLAAC_X=1
run "round 3's forms" $OLD
run "default" PLAAC_X=1
run "round 3's forms" $OLD
run "default" PLAAC_X=1
run "default, 3 runs" PLAAC_TRACK_SEGMENTS=3
cat $out
