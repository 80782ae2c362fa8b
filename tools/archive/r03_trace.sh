#!/bin/bash
# kernel timelines (rocprofv3 --kernel-trace) of one step of the working tree: bench args as given
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r3; mkdir -p $O
export PLAAC_STREAM_DEBUG=1
A="--no-e2e --no-cpu-baseline --steps 4 --warmup 2"
rm -rf $O/tn
rocprofv3 --kernel-trace --output-format csv -d $O/tn -- python3 bench.py $A "$@" > $O/tn.json 2> $O/tn.err
grep "^plaac:" $O/tn.err
python3 tools/timeline.py $(find $O/tn -name "*kernel_trace.csv" | head -1) 2 | tee $O/timeline_tn.txt
rm -rf $O/tn
