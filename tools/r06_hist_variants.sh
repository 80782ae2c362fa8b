#!/bin/bash
# the background pass of the 10 M-sequence batch timed with several builds of the library in turn (two rounds), one gpurun
# call.  usage: LIBS="a=build/liba.so b=build/libb.so" TAG=x bash tools/r06_hist_variants.sh
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/r6
OUT=gpurun_out/r6/hist_variants_${TAG:-x}.txt
: > $OUT
one() { # label, lib
    PLAAC_NATIVE_LIB=$2 timeout -k 10 300 python3 bench.py --steps 3 --no-cpu-baseline --no-e2e --no-predict --no-tracks-leg --no-host-leg --no-clock-probe --no-tolerance-leg 2> gpurun_out/r6/hist_ab.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); h=d['roofline']['histogram_pass']
print('$1', 'k_hist', h['ms'], 'ms', h.get('counts_sha16'), h['frac_of_peak'], 'of peak')" | tee -a $OUT
}
for rep in 1 2; do
    for kv in $LIBS; do one ${kv%%=*} ${kv#*=} || exit 1; done
done
