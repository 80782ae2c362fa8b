#!/bin/bash
# tools/r04_trace.sh <tag> [nsteps] [bench args] - kernel timeline (rocprofv3 --kernel-trace) of steady-state steps
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4; mkdir -p $O
TAG=$1; NS=$2; shift 2
rm -rf $O/tn
rocprofv3 --kernel-trace --output-format csv -d $O/tn -- python3 bench.py --no-e2e --no-cpu-baseline --no-clock-probe --no-host-leg --steps 8 --warmup 3 "$@" > $O/tn.json 2> $O/tn.err
python3 tools/timeline.py $(find $O/tn -name "*kernel_trace.csv" | head -1) $((NS + 12)) $NS > $O/timeline_$TAG.txt
rm -rf $O/tn
