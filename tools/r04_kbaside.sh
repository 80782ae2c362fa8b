#!/bin/bash
# window-track kernels of overlapping calls off the caller's stream (PLAAC_KB_ASIDE): same-box A/B
mkdir -p gpurun_out/r4
out=gpurun_out/r4/kbaside.txt
: > $out
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu  > gpurun_out/r4/kbaside_pytest.txt 2>&1 || { tail -30 gpurun_out/r4/kbaside_pytest.txt; exit 1; }
tail -2 gpurun_out/r4/kbaside_pytest.txt
F="--steps 16 --warmup 4 --no-cpu-baseline --no-e2e --no-predict --no-clock-probe --no-host-leg"
run() { # label, env...
  echo "== $1" >> $out; shift
  env "$@" timeout -k 10 300 python3 bench.py $F 2>>gpurun_out/r4/kbaside.err | python3 -c "
import sys,json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print('   ms_per_step', d['ms_per_step'], 'no_overlap', (d.get('config') or {}).get('no_overlap_ms_per_step'), 'match', (d.get('cpu_baseline') or {}).get('gpu_rows_match_oracle'))
" >> $out || echo "   failed" >> $out
}
run "default" PLAAC_X=1
run "default" PLAAC_X=1
run "calibration: finish kernel, prio 0/0" PLAAC_FINISH_KERNEL=1 PLAAC_KB_PRIO=00
cat $out
