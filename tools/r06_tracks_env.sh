#!/bin/bash
# round 6: track mode, same box: environment variants of the tree's library, alternating, three rounds.
#   tools/r06_tracks_env.sh "label1:VAR=v,VAR2=w" "label2:" ...      (TAG=name for the output file; SKIP_TESTS=1)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r6; mkdir -p $O; TAG=${TAG:-env}
if [ "${SKIP_TESTS:-0}" != 1 ]; then
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "${TESTS:-track}" > $O/${TAG}_tests.txt 2>&1; tail -n 5 $O/${TAG}_tests.txt
grep -q "passed" $O/${TAG}_tests.txt && ! grep -q "failed" $O/${TAG}_tests.txt || exit 1
fi
out=$O/tracks_${TAG}.txt; : > $out
F="--steps 16 --warmup 4 --no-cpu-baseline --no-e2e --no-predict --no-clock-probe --no-host-leg --no-tracks-leg --no-tolerance-leg ${BENCH_ARGS:---tracks --nprot 1250000}"
for rep in 1 2 3; do
  for spec in "$@"; do
    L=${spec%%:*}; E=${spec#*:}
    env X_=1 ${E//,/ } timeout -k 10 300 python3 bench.py $F 2>>$O/tracks_${TAG}.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-14s' % '$L', 'ms/step', d['ms_per_step'], {k: round(x,3) for k,x in d['roofline']['kernel_ms'].items()})" >> $out || echo "$L failed" >> $out
  done
done
cat $out
