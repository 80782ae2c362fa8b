#!/bin/bash
# round 5: the listed proteins' path words in runs of their own (k_vit<LIST> -> k_core_list): parity, then same-box A/B against
# the library built from the commit before (plaac_amd/libplaac_native_base.so, built by hand: git stash; make; cp; git stash pop)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r5; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > $O/cwords_tests.log 2>&1 || { tail -20 $O/cwords_tests.log; exit 1; }
tail -1 $O/cwords_tests.log
Q="--no-e2e --no-predict --no-tracks-leg --no-cpu-baseline --no-host-leg --no-clock-probe"
{
for rep in 1 2 3; do
  for lib in base new; do
    for cfg in "" "--nprot 1250000" "--sweep --nprot 1250000"; do
      L=$GRAFT_REPO_ROOT/plaac_amd/libplaac_native.so; [ $lib = base ] && L=$GRAFT_REPO_ROOT/plaac_amd/libplaac_native_base.so
      PLAAC_NATIVE_LIB=$L python3 bench.py $cfg $Q --steps 20 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-5s %-28s %8.4f ms per step' % ('$lib', '$cfg' or '(cfg4, 10 M)', d['ms_per_step']))"
    done
  done
done
} > $O/ab_cwords.txt 2>&1
cat $O/ab_cwords.txt
