#!/bin/bash
# round 5: the library built with another instruction-scheduling strategy of the compiler (-mllvm -amdgpu-sched-strategy=...)
# against the default build, same box, alternating: parity under the variant, then the bench lines
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r5; mkdir -p $O
V=${1:-max-ilp}
PLAAC_NATIVE_LIB=$GRAFT_REPO_ROOT/plaac_amd/libplaac_native_$V.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "config or real_proteome or adversarial or tracks" > $O/sched_tests.log 2>&1 || { tail -20 $O/sched_tests.log; exit 1; }
tail -1 $O/sched_tests.log
Q="--no-e2e --no-predict --no-tracks-leg --no-cpu-baseline --no-host-leg --no-clock-probe"
{
for rep in 1 2 3; do
  for lib in base $V; do
    for cfg in "" "--tracks --nprot 1250000" "--config 3"; do
      L=$GRAFT_REPO_ROOT/plaac_amd/libplaac_native.so; [ $lib = base ] || L=$GRAFT_REPO_ROOT/plaac_amd/libplaac_native_$V.so
      PLAAC_NATIVE_LIB=$L python3 bench.py $cfg $Q --steps 20 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-8s %-28s %8.4f ms per step' % ('$lib', '$cfg' or '(cfg4, 10 M)', d['ms_per_step']))"
    done
  done
done
} > $O/ab_sched_$V.txt 2>&1
cat $O/ab_sched_$V.txt
