#!/bin/bash
# tools/ab.sh VAR "v1 v2 ..." [bench args] - same-box A/B of an environment knob (boxes differ by +-10 %, so only
# comparisons inside one gpurun call count): the bench step for every value, twice, alternating.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
VAR=$1; VALS=$2; shift 2
for rep in 1 2; do for v in $VALS; do
  env $VAR=$v timeout 300 python3 bench.py --no-e2e --no-cpu-baseline --steps 20 "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$VAR=$v', '$*', 'ms/step', d['ms_per_step'], {k: round(x,3) for k,x in d['roofline']['kernel_ms'].items()})"
done; done
