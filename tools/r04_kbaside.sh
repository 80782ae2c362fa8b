#!/bin/bash
# window-track kernels of overlapping calls off the caller's stream (PLAAC_KB_ASIDE): same-box A/B
mkdir -p gpurun_out/r4
out=gpurun_out/r4/kbaside.txt
: > $out
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu  > gpurun_out/r4/kbaside_pytest.txt 2>&1 || { tail -30 gpurun_out/r4/kbaside_pytest.txt; exit 1; }
tail -2 gpurun_out/r4/kbaside_pytest.txt
F="--steps 40 --warmup 6 --no-cpu-baseline --no-e2e --no-predict --no-clock-probe --no-host-leg --nprot 1250000"
run() { # label, env...
  echo "== $1" >> $out; shift
  env "$@" timeout -k 10 300 python3 bench.py $F 2>>gpurun_out/r4/kbaside.err | python3 -c "
import sys,json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print('   ms_per_step', d['ms_per_step'])
" >> $out || echo "   failed" >> $out
}
run "share: finish kernel for the rest" PLAAC_FINISH_KERNEL=1
run "share: rest writes its rows" PLAAC_X=1
run "share: finish kernel for the rest" PLAAC_FINISH_KERNEL=1
run "share: rest writes its rows" PLAAC_X=1
F="--steps 40 --warmup 6 --no-cpu-baseline --no-e2e --no-predict --no-clock-probe --no-host-leg --config 3"
run "cfg3" PLAAC_X=1
F="--steps 40 --warmup 6 --no-cpu-baseline --no-e2e --no-predict --no-clock-probe --no-host-leg --config 2"
run "cfg2" PLAAC_X=1
cat $out
