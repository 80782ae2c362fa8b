#!/bin/bash
# round 5: several whole-library scheduling variants against the default build, headline + tracks only, more repetitions
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r5; mkdir -p $O
Q="--no-e2e --no-predict --no-tracks-leg --no-cpu-baseline --no-host-leg --no-clock-probe"
{
for rep in 1 2 3 4; do
  for lib in base $*; do
    for cfg in "" "--tracks --nprot 1250000"; do
      L=$GRAFT_REPO_ROOT/plaac_amd/libplaac_native.so; [ $lib = base ] || L=$GRAFT_REPO_ROOT/plaac_amd/libplaac_native_$lib.so
      PLAAC_NATIVE_LIB=$L python3 bench.py $cfg $Q --steps 20 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-18s %-28s %8.4f ms per step' % ('$lib', '$cfg' or '(cfg4, 10 M)', d['ms_per_step']))"
    done
  done
done
} > $O/ab_sched2.txt 2>&1
sort -k2,3 -s $O/ab_sched2.txt
