#!/bin/bash
# round 5: track mode, same box: the forward pass with its posterior stores lane by lane (round 4) against transposed through LDS
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r5; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_real_proteomes.py -x -q -m gpu -k "track" > $O/ab4_tests.txt 2>&1; tail -n 5 $O/ab4_tests.txt
grep -q "passed" $O/ab4_tests.txt && ! grep -q "failed" $O/ab4_tests.txt || exit 1
out=$O/tracks_ab4.txt; : > $out
F="--steps 16 --warmup 4 --no-cpu-baseline --no-e2e --no-predict --no-clock-probe --no-host-leg --no-tracks-leg --tracks --nprot 1250000"
run() { L=$1; shift
  env "$@" timeout -k 10 300 python3 bench.py $F 2>>$O/tracks_ab4.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L:', 'ms/step', d['ms_per_step'], d['roofline']['kernel_ms'])" >> $out || echo "$L failed" >> $out
}
for rep in 1 2 3; do
  run "r04 forms                      " PLAAC_TRACK_POST_FORM=0 PLAAC_TRACK_POST_OCC=2 PLAAC_TRACK_KB_LATE=0 PLAAC_TRACK_VIT_EARLY=0
  run "lane stores, occ 3, vit early  " PLAAC_TRACK_POST_FORM=0 PLAAC_TRACK_KB_LATE=0
  run "transposed 256, kb early       " PLAAC_TRACK_POST_FORM=1 PLAAC_TRACK_KB_LATE=0
  run "transposed 256, kb late        " PLAAC_TRACK_POST_FORM=1 PLAAC_TRACK_KB_LATE=1
  run "transposed 768, kb early       " PLAAC_TRACK_POST_FORM=2 PLAAC_TRACK_KB_LATE=0
  run "transposed 768, kb late        " PLAAC_TRACK_POST_FORM=2 PLAAC_TRACK_KB_LATE=1
done
cat $out
rm -rf $O/tn
PLAAC_TRACK_KB_LATE=0 rocprofv3 --kernel-trace --output-format csv -d $O/tn -- python3 bench.py $F --steps 4 --warmup 2 > $O/tn_tracks.json 2> $O/tn.err
python3 tools/timeline.py $(find $O/tn -name "*kernel_trace.csv" | head -1) 2 | grep -v vectorized > $O/timeline_tracks_transposed.txt
rm -rf $O/tn
cut -c1-100 $O/timeline_tracks_transposed.txt
