#!/bin/bash
# tools/ab_lib.sh OLD.so [bench args] - same-box A/B of two builds of the library (PLAAC_NATIVE_LIB selects the .so):
# the bench step with the old build and with the tree's, twice, alternating.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OLD=$GRAFT_REPO_ROOT/$1; shift
for rep in 1 2; do for v in old new; do
  if [ $v = old ]; then export PLAAC_NATIVE_LIB=$OLD; else unset PLAAC_NATIVE_LIB; fi
  timeout 300 python3 bench.py --no-e2e --no-cpu-baseline --no-clock-probe --steps 20 "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', '$*', 'ms/step', d['ms_per_step'], {k: round(x,3) for k,x in d['roofline']['kernel_ms'].items()})"
done; done
