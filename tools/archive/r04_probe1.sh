#!/bin/bash
# tools/r04_probe1.sh - round-4 opening measurements on one box: the 1.25 M-sequence share at HEAD, by itself / overlapping
# steps / two contexts alternating, under 4 / 12 / 24 hardware queues (GPU_MAX_HW_QUEUES did not exist as a lever when
# "--contexts 2: no gain" was recorded).
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r4; mkdir -p $O
line() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'ms/step', d['ms_per_step'], 'alone', (d['config'].get('step_by_itself') or {}).get('ms_per_step'), {k: round(x,3) for k,x in d['roofline']['kernel_ms'].items()})"; }
for rep in 1 2; do
for q in 0 12 24; do
  for c in 1 2; do
    if [ $q = 0 ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
    timeout 300 python3 bench.py --nprot 1250000 --contexts $c --no-e2e --no-cpu-baseline --no-clock-probe --no-host-leg --steps 40 2>/dev/null | line "queues=$q contexts=$c"
  done
done
done
unset GPU_MAX_HW_QUEUES
timeout 300 python3 bench.py --no-e2e --no-cpu-baseline --no-clock-probe --no-host-leg --steps 20 2>/dev/null | line "cfg4 full"
