#!/bin/bash
# tools/serial_ab.sh [OLD_LIB=path] - kernels alone (PLAAC_SERIAL_STREAMS=1) for an old build of the library and the tree's, track mode, same box
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for v in old new; do
  if [ $v = old ]; then export PLAAC_NATIVE_LIB=$GRAFT_REPO_ROOT/${OLD_LIB:-tools_tmp/old_lib.so}; else unset PLAAC_NATIVE_LIB; fi
  PLAAC_SERIAL_STREAMS=1 timeout 300 python3 bench.py --no-e2e --no-cpu-baseline --no-clock-probe --steps 8 --tracks 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v serial', 'ms/step', d['ms_per_step'], {k: round(x,3) for k,x in d['roofline']['kernel_ms'].items()})"
done
