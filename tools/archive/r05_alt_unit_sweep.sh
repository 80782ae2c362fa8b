#!/bin/bash
# round 5: the sweep over the share (chain-bound: its step ends with the throughput-form chains of one 36,000-residue protein)
# with throughput-form chain kernels from the max-ilp unit (tools/experiments/r05_alt_unit_throughput_kernels.patch applied)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r5; mkdir -p $O
Q="--no-e2e --no-predict --no-tracks-leg --no-cpu-baseline --no-host-leg --no-clock-probe"
{
for rep in 1 2 3; do
  for mask in 0 2 4 6 30; do
    PLAAC_ALT_UNIT=$mask python3 bench.py --sweep --nprot 1250000 $Q --steps 10 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('PLAAC_ALT_UNIT=%-3s --sweep --nprot 1250000 %8.4f ms per step' % ('$mask', d['ms_per_step']))"
  done
done
} > $O/ab_alt_unit_sweep.txt 2>&1
cat $O/ab_alt_unit_sweep.txt
