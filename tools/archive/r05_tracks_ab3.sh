#!/bin/bash
# round 5: track mode, same box: one-pass kernel at 3 waves per SIMD beside a window kernel held to fewer blocks per CU
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r5; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "track_mode_posteriors" > $O/ab3_tests.txt 2>&1; tail -n 3 $O/ab3_tests.txt
grep -q "passed" $O/ab3_tests.txt && ! grep -q "failed" $O/ab3_tests.txt || exit 1
out=$O/tracks_ab3.txt; : > $out
F="--steps 16 --warmup 4 --no-cpu-baseline --no-e2e --no-predict --no-clock-probe --no-host-leg --no-tracks-leg --tracks --nprot 1250000"
run() { L=$1; shift
  env "$@" timeout -k 10 300 python3 bench.py $F 2>>$O/tracks_ab3.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L:', 'ms/step', d['ms_per_step'], d['roofline']['kernel_ms'])" >> $out || echo "$L failed" >> $out
}
for rep in 1 2; do
  run "r04 forms                    " PLAAC_TRACK_ONE_PASS=0 PLAAC_TRACK_KB_LATE=0 PLAAC_TRACK_VIT_EARLY=0
  run "one pass, late, pad 0        " PLAAC_TRACK_KB_LDS=0
  run "one pass, late, pad 2048 (9) " PLAAC_TRACK_KB_LDS=2048
  run "one pass, late, pad 4608 (8) " PLAAC_TRACK_KB_LDS=4608
  run "one pass, late, pad 7400 (7) " PLAAC_TRACK_KB_LDS=7400
  run "one pass, late, pad 11000 (6)" PLAAC_TRACK_KB_LDS=11000
  run "one pass, early, pad 4608 (8)" PLAAC_TRACK_KB_LDS=4608 PLAAC_TRACK_KB_LATE=0
  run "two kernels, late, pad 4608  " PLAAC_TRACK_KB_LDS=4608 PLAAC_TRACK_ONE_PASS=0
done
cat $out
rm -rf $O/tn
PLAAC_TRACK_KB_LDS=4608 rocprofv3 --kernel-trace --output-format csv -d $O/tn -- python3 bench.py $F --steps 4 --warmup 2 > $O/tn_tracks.json 2> $O/tn.err
python3 tools/timeline.py $(find $O/tn -name "*kernel_trace.csv" | head -1) 2 | grep -v vectorized > $O/timeline_tracks_one_pass_pad8.txt
rm -rf $O/tn
cut -c1-100 $O/timeline_tracks_one_pass_pad8.txt
