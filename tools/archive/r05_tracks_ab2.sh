#!/bin/bash
# round 5: track mode, same box: one-pass (k_bwd_fwd_post) vs two kernels, window kernel early / late; parity first; trace of the default
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r5; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_real_proteomes.py -x -q -m gpu -k "track" > $O/ab2_tests.txt 2>&1; tail -n 3 $O/ab2_tests.txt
grep -q "passed" $O/ab2_tests.txt && ! grep -q "failed" $O/ab2_tests.txt || exit 1
out=$O/tracks_ab2.txt; : > $out
F="--steps 16 --warmup 4 --no-cpu-baseline --no-e2e --no-predict --no-clock-probe --no-host-leg --no-tracks-leg --tracks --nprot 1250000"
run() { L=$1; shift
  env "$@" timeout -k 10 300 python3 bench.py $F 2>>$O/tracks_ab2.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L:', 'ms/step', d['ms_per_step'], d['roofline']['kernel_ms'])" >> $out || echo "$L failed" >> $out
}
for rep in 1 2 3; do
  run "r04 forms            " PLAAC_TRACK_ONE_PASS=0 PLAAC_TRACK_KB_LATE=0 PLAAC_TRACK_VIT_EARLY=0
  run "one pass, kb early   " PLAAC_TRACK_ONE_PASS=1 PLAAC_TRACK_KB_LATE=0
  run "one pass, kb late    " PLAAC_TRACK_ONE_PASS=1 PLAAC_TRACK_KB_LATE=1
done
cat $out
rm -rf $O/tn
rocprofv3 --kernel-trace --output-format csv -d $O/tn -- python3 bench.py $F --steps 4 --warmup 2 > $O/tn_tracks.json 2> $O/tn.err
python3 tools/timeline.py $(find $O/tn -name "*kernel_trace.csv" | head -1) 2 | grep -v vectorized > $O/timeline_tracks_one_pass.txt
rm -rf $O/tn
cut -c1-100 $O/timeline_tracks_one_pass.txt
