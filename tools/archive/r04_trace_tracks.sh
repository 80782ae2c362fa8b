#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4; mkdir -p $O
rm -rf $O/tn
rocprofv3 --kernel-trace --output-format csv -d $O/tn -- python3 bench.py --no-e2e --no-cpu-baseline --no-clock-probe --no-host-leg --no-predict --steps 4 --warmup 2 --tracks --nprot 1250000 > $O/tn_tracks.json 2> $O/tn.err
python3 tools/timeline.py $(find $O/tn -name "*kernel_trace.csv" | head -1) 2 > $O/timeline_tracks_new.txt
rm -rf $O/tn
cat $O/timeline_tracks_new.txt | cut -c1-100
