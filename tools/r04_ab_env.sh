#!/bin/bash
# tools/r04_ab_env.sh VAR "v1 v2" - GPU suite, then same-box A/B of an environment knob on the full batch and the share
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r4
VAR=$1; VALS=$2
timeout -k 10 600 python3 -m pytest tests -x -q -m gpu --timeout=200 > $O/ab_suite.txt 2>&1; tail -4 $O/ab_suite.txt
line() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'ms/step', d['ms_per_step'], 'alone', (d['config'].get('step_by_itself') or {}).get('ms_per_step'), {k: round(x,3) for k,x in d['roofline']['kernel_ms'].items()}, (d.get('cpu_baseline') or {}).get('gpu_rows_match_oracle'))"; }
for rep in 1 2; do for v in $VALS; do
  env $VAR=$v timeout 300 python3 bench.py --no-e2e --no-cpu-baseline --no-clock-probe --no-host-leg --no-predict --steps 20 2>/dev/null | line "full $VAR=$v"
  env $VAR=$v timeout 300 python3 bench.py --nprot 1250000 --no-e2e --no-cpu-baseline --no-clock-probe --no-host-leg --steps 40 2>/dev/null | line "share $VAR=$v"
done; done
timeout 300 python3 bench.py --no-e2e --no-clock-probe --no-host-leg --no-predict --steps 20 2>/dev/null | line "full checked"
