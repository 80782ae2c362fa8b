#!/bin/bash
# round 5: the latency-form chain kernels from the unit built with max-ilp scheduling (default) against the main unit's copies
# (PLAAC_LAT_UNIT=0): parity, then same-box A/B of the chain-bound lines and the headline
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r5; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_real_proteomes.py -x -q -m gpu > $O/lat_unit_tests.log 2>&1 || { tail -20 $O/lat_unit_tests.log; exit 1; }
tail -1 $O/lat_unit_tests.log
Q="--no-e2e --no-predict --no-tracks-leg --no-cpu-baseline --no-host-leg --no-clock-probe"
{
for rep in 1 2 3; do
  for mode in 0 1; do
    for cfg in "--config 3" "--config 2" "--nprot 1250000 --no-overlap" "--nprot 1250000" "--tracks --config 3" ""; do
      PLAAC_LAT_UNIT=$mode python3 bench.py $cfg $Q --steps 20 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('PLAAC_LAT_UNIT=$mode  %-30s %8.4f ms per step' % ('$cfg' or '(cfg4, 10 M)', d['ms_per_step']))"
    done
  done
done
} > $O/ab_lat_unit.txt 2>&1
cat $O/ab_lat_unit.txt
