#!/bin/bash
# GPU suite + short bench lines (full batch, share, cfg2, cfg3) at the working tree
mkdir -p gpurun_out/r4
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > gpurun_out/r4/check_pytest.txt 2>&1; rc=$?
tail -5 gpurun_out/r4/check_pytest.txt
[ $rc -eq 0 ] || exit 1
F="--steps 16 --warmup 4 --no-e2e --no-predict --no-clock-probe --no-host-leg"
out=gpurun_out/r4/check_lines.txt; : > $out
line() { echo "== $*" >> $out; timeout -k 10 400 python3 bench.py --allow-diagnostics $F "$@" 2>>gpurun_out/r4/check.err | python3 -c "
import sys,json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print('   ms_per_step', d['ms_per_step'], 'match', (d.get('cpu_baseline') or {}).get('gpu_rows_match_oracle'))
" >> $out || echo '   failed' >> $out; }
line
line --nprot 1250000 --no-cpu-baseline
line --config 2 --no-cpu-baseline
line --config 3 --no-cpu-baseline
PLAAC_NATIVE_LIB=$PWD/plaac_amd/libplaac_native_diag.so PLAAC_DEBUG_SKIP=k_core_list line --no-cpu-baseline
cat $out
