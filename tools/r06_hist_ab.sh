#!/bin/bash
# k_hist A/B inside one gpurun call: the histogram parity tests with the tree's library, then the background pass of the
# 10 M-sequence batch timed with OLD_LIB and with the tree's library in turn (bench.py's histogram_pass: 3 launches by HIP
# events).  usage: OLD_LIB=build/libplaac_hist_old.so TAG=v1 bash tools/r06_hist_ab.sh
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/r6
OUT=gpurun_out/r6/hist_ab_${TAG:-x}.txt
: > $OUT
timeout -k 10 400 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_dist.py -x -q -m gpu -k "hist" > gpurun_out/r6/hist_tests.txt 2>&1
echo "tests rc=$? $(tail -n 1 gpurun_out/r6/hist_tests.txt)" | tee -a $OUT
grep -q passed gpurun_out/r6/hist_tests.txt && ! grep -q failed gpurun_out/r6/hist_tests.txt || { tail -n 30 gpurun_out/r6/hist_tests.txt; exit 1; }
one() { # label, lib
    PLAAC_NATIVE_LIB=$2 timeout -k 10 300 python3 bench.py --steps 5 --no-cpu-baseline --no-e2e --no-predict --no-tracks-leg --no-host-leg --no-clock-probe --no-tolerance-leg 2> gpurun_out/r6/hist_ab.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); h=d['roofline']['histogram_pass']
print('$1', 'k_hist', h['ms'], 'ms', h.get('counts_sha16'), h['frac_of_peak'], 'of peak; step', d['ms_per_step'], 'ms; oracle check', d.get('oracle_check'))" | tee -a $OUT
}
for rep in 1 2; do
    [ -n "$OLD_LIB" ] && one old $OLD_LIB
    one new plaac_amd/libplaac_native.so
done
