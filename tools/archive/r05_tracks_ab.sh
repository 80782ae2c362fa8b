#!/bin/bash
# round 5: track mode (1.25 M-sequence share), same box: old launch order vs window kernel behind the packed copy vs + Viterbi bytes early;
# then a kernel trace of the new default. Output: gpurun_out/r5/tracks_ab.txt, timeline_tracks.txt
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r5; mkdir -p $O
out=$O/tracks_ab.txt; : > $out
F="--steps 16 --warmup 4 --no-cpu-baseline --no-e2e --no-predict --no-clock-probe --no-host-leg --no-tracks-leg --tracks --nprot 1250000"
run() { # label, env...
  L=$1; shift
  env "$@" timeout -k 10 300 python3 bench.py $F 2>>$O/tracks_ab.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L:', 'ms/step', d['ms_per_step'], d['roofline']['kernel_ms'])" >> $out || echo "$L failed" >> $out
}
for rep in 1 2 3; do
  run "r04 order          " PLAAC_TRACK_KB_LATE=0 PLAAC_TRACK_VIT_EARLY=0
  run "kb late            " PLAAC_TRACK_KB_LATE=1 PLAAC_TRACK_VIT_EARLY=0
  run "kb late + vit early" PLAAC_TRACK_KB_LATE=1 PLAAC_TRACK_VIT_EARLY=1
  run "vit early only     " PLAAC_TRACK_KB_LATE=0 PLAAC_TRACK_VIT_EARLY=1
done
cat $out
rm -rf $O/tn
rocprofv3 --kernel-trace --output-format csv -d $O/tn -- python3 bench.py $F --steps 4 --warmup 2 > $O/tn_tracks.json 2> $O/tn.err
python3 tools/timeline.py $(find $O/tn -name "*kernel_trace.csv" | head -1) 2 > $O/timeline_tracks.txt
rm -rf $O/tn
cut -c1-100 $O/timeline_tracks.txt
