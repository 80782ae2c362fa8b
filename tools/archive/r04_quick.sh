#!/bin/bash
# tools/r04_quick.sh - GPU suite, the share / cfg2 / cfg3 / full lines, a timeline of the share
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -5
PLAAC_MIXED_GROUPS=3 PLAAC_MIXED_MIN_REST=1 timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_real_proteomes.py -x -q -m gpu 2>&1 | tail -3
line() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'ms/step', d['ms_per_step'], 'alone', (d['config'].get('step_by_itself') or {}).get('ms_per_step'), {k: round(x,3) for k,x in d['roofline']['kernel_ms'].items()}, (d.get('cpu_baseline') or {}).get('gpu_rows_match_oracle'))"; }
for rep in 1 2; do
  timeout 300 python3 bench.py --nprot 1250000 --no-e2e --no-cpu-baseline --no-clock-probe --no-host-leg --steps 40 2>/dev/null | line "share"
done
timeout 300 python3 bench.py --config 2 --steps 200 --no-e2e --no-cpu-baseline --no-clock-probe --no-host-leg 2>/dev/null | line "cfg2"
timeout 300 python3 bench.py --config 3 --steps 100 --no-e2e --no-cpu-baseline --no-clock-probe --no-host-leg 2>/dev/null | line "cfg3"
timeout 300 python3 bench.py --nprot 1250000 --no-e2e --no-clock-probe --no-host-leg --steps 40 2>/dev/null | line "share checked"
timeout 300 python3 bench.py --no-e2e --no-cpu-baseline --no-clock-probe --no-host-leg --steps 20 2>/dev/null | line "cfg4 full"
bash tools/r04_trace.sh share 3 --nprot 1250000
