#!/bin/bash
mkdir -p gpurun_out/r4
out=gpurun_out/r4/tracks_ab.txt
: > $out
F="--steps 10 --warmup 3 --no-cpu-baseline --no-e2e --no-predict --no-clock-probe --no-host-leg --tracks --nprot 1250000"
run() { echo "== $1" >> $out; shift
  env "$@" timeout -k 10 300 python3 bench.py $F 2>>gpurun_out/r4/tracks_ab.err | python3 -c "
import sys,json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print('   ms_per_step', d['ms_per_step'])
" >> $out || echo "   failed" >> $out; }
run "default" PLAAC_X=1
run "throughput forms" PLAAC_LATENCY_MODE=0
run "throughput forms, 3 runs" PLAAC_LATENCY_MODE=0 PLAAC_TRACK_SEGMENTS=3
run "throughput forms, 4 runs" PLAAC_LATENCY_MODE=0 PLAAC_TRACK_SEGMENTS=4
run "latency forms, 3 runs" PLAAC_TRACK_SEGMENTS=3
run "latency forms, 4 runs" PLAAC_TRACK_SEGMENTS=4
run "throughput forms, 1 run" PLAAC_LATENCY_MODE=0 PLAAC_TRACK_SEGMENTS=1
run "default" PLAAC_X=1
cat $out
