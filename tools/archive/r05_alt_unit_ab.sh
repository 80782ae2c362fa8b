#!/bin/bash
# round 5: which throughput-form chain kernels gain INSIDE the concurrent step when they come from the unit built with max-ilp
# scheduling (PLAAC_ALT_UNIT mask: 2 k_fwd, 4 k_win, 8 k_vit list form, 16 k_core_list)? Alone every one of them is faster
# (rocprofv3, streams serialised: k_fwd -19 %, k_win -14 %, k_vit -9 %, k_core_list -9 %); the whole library under max-ilp is not.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r5; mkdir -p $O
Q="--no-e2e --no-predict --no-tracks-leg --no-cpu-baseline --no-host-leg --no-clock-probe"
{
for rep in 1 2 3; do
  for mask in ${MASKS:-0 2 4 8 16 30}; do
    for cfg in "" "--nprot 1250000"; do
      PLAAC_ALT_UNIT=$mask python3 bench.py $cfg $Q --steps 20 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('PLAAC_ALT_UNIT=%-3s %-20s %8.4f ms per step' % ('$mask', '$cfg' or '(cfg4, 10 M)', d['ms_per_step']))"
    done
  done
done
} > $O/ab_alt_unit.txt 2>&1
cat $O/ab_alt_unit.txt
