#!/bin/bash
# round 6: track mode, same box: an older build of the library (OLD_LIB=build/libplaac_r05.so) against the tree's -
# the track-mode parity tests first, then the step alternating, then the kernels by themselves (serial streams)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r6; mkdir -p $O; TAG=${TAG:-ab}
if [ "${SKIP_TESTS:-0}" != 1 ]; then
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "${TESTS:-track}" > $O/${TAG}_tests.txt 2>&1; tail -n 5 $O/${TAG}_tests.txt
grep -q "passed" $O/${TAG}_tests.txt && ! grep -q "failed" $O/${TAG}_tests.txt || exit 1
fi
out=$O/tracks_${TAG}.txt; : > $out
F="--steps 16 --warmup 4 --no-cpu-baseline --no-e2e --no-predict --no-clock-probe --no-host-leg --no-tracks-leg --tracks --nprot 1250000"
run() { L=$1; shift
  env "$@" timeout -k 10 300 python3 bench.py $F 2>>$O/tracks_${TAG}.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L:', 'ms/step', d['ms_per_step'], {k: round(x,3) for k,x in d['roofline']['kernel_ms'].items()})" >> $out || echo "$L failed" >> $out
}
OLD=$GRAFT_REPO_ROOT/${OLD_LIB:-build/libplaac_r05.so}
for rep in 1 2 3; do
  run "old        " PLAAC_NATIVE_LIB=$OLD
  run "new        " X=1
done
run "old serial " PLAAC_NATIVE_LIB=$OLD PLAAC_SERIAL_STREAMS=1
run "new serial " PLAAC_SERIAL_STREAMS=1
cat $out
