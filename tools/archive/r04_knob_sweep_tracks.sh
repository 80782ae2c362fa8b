#!/bin/bash
# track mode over the 1.25 M-sequence share: knob settings one after the other on one box (default first, in the middle and last)
mkdir -p gpurun_out/r4
out=gpurun_out/r4/knob_sweep_tracks.txt
: > $out
F="--steps 10 --warmup 3 --no-cpu-baseline --no-e2e --no-predict --no-clock-probe --no-host-leg --tracks --nprot 1250000"
run() { echo "== $1" >> $out; shift
  env "$@" timeout -k 10 300 python3 bench.py $F 2>>gpurun_out/r4/knob_sweep_tracks.err | python3 -c "
import sys,json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print('   ms_per_step', d['ms_per_step'])
" >> $out || echo "   failed" >> $out; }
run "default" PLAAC_X=1
run "1 run" PLAAC_TRACK_SEGMENTS=1
run "3 runs" PLAAC_TRACK_SEGMENTS=3
run "4 runs" PLAAC_TRACK_SEGMENTS=4
run "default" PLAAC_X=1
run "long-protein blocks 256" PLAAC_TRACK_CONSEC=256
run "long-protein blocks 8192" PLAAC_TRACK_CONSEC=8192
run "k_vit unmixed" PLAAC_TRACK_VIT_MIXED=0
run "12 hw queues" GPU_MAX_HW_QUEUES=12
run "default" PLAAC_X=1
cat $out
