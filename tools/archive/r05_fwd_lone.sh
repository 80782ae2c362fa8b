#!/bin/bash
# round 5, VERDICT r04 #7: the paired forward chain - PLAAC_FWD_LONE=1: a lone pair's table entries through the scalar cache,
# 2: one ds_read2_b64 instead of two ds_read_b64, 3: that with the step's instruction order fixed by hand - against the
# default (0): parity under each switch, then same-box A/B of the chain-bound lines
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r5; mkdir -p $O
for m in 1 2 3; do
  PLAAC_FWD_LONE=$m timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "config or long or chain or latency or adversarial or real_proteome" > $O/fwd_lone_tests_$m.log 2>&1 || { tail -20 $O/fwd_lone_tests_$m.log; exit 1; }
  echo "PLAAC_FWD_LONE=$m: $(tail -1 $O/fwd_lone_tests_$m.log)"
done
Q="--no-e2e --no-predict --no-tracks-leg --no-cpu-baseline --no-host-leg --no-clock-probe"
{
for rep in 1 2 3; do
  for mode in 0 1 2 3; do
    for cfg in "--config 3" "--config 2" "--nprot 1250000 --no-overlap"; do
      PLAAC_FWD_LONE=$mode python3 bench.py $cfg $Q --steps 20 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('PLAAC_FWD_LONE=$mode  %-30s %8.4f ms per step' % ('$cfg', d['ms_per_step']))"
    done
  done
done
} > $O/ab_forward_lut.txt 2>&1
cat $O/ab_forward_lut.txt
