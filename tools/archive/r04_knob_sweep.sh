#!/bin/bash
mkdir -p gpurun_out/r4
out=gpurun_out/r4/knob_sweep.txt
: > $out
F="--steps 16 --warmup 4 --no-cpu-baseline --no-e2e --no-predict --no-clock-probe --no-host-leg"
run() { echo "== $1" >> $out; shift
  env "$@" timeout -k 10 300 python3 bench.py $F 2>>gpurun_out/r4/knob_sweep.err | python3 -c "
import sys,json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print('   ms_per_step', d['ms_per_step'])
" >> $out || echo "   failed" >> $out; }
run "default" PLAAC_X=1
run "refine grid 6/CU" PLAAC_RF_GRID=1755
run "refine grid 5/CU" PLAAC_RF_GRID=1462
run "refine grid 4/CU" PLAAC_RF_GRID=1170
run "2 runs" PLAAC_PIPE_SEGMENTS=2
run "4 runs" PLAAC_PIPE_SEGMENTS=4
run "default" PLAAC_X=1
run "core list off" PLAAC_CORE_LIST=0
run "12 hw queues" GPU_MAX_HW_QUEUES=12
cat $out
