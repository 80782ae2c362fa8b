#!/bin/bash
# round 5: kernel timeline of one 9-point sweep call over the 1.25 M share (GPU_MAX_HW_QUEUES=12 as bench.py --sweep sets it)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r5; mkdir -p $O; rm -rf $O/ts
F="--no-cpu-baseline --no-e2e --no-predict --no-clock-probe --no-host-leg --no-tracks-leg --sweep --nprot 1250000"
rocprofv3 --kernel-trace --output-format csv -d $O/ts -- python3 bench.py $F --steps 4 --warmup 2 "$@" > $O/ts.json 2> $O/ts.err
python3 tools/timeline.py $(find $O/ts -name "*kernel_trace.csv" | head -1) 2 | grep -v vectorized > $O/timeline_sweep_share.txt
rm -rf $O/ts
cut -c1-110 $O/timeline_sweep_share.txt
