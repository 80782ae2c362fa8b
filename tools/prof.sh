#!/bin/bash
# tools/prof.sh [bench args]: development profile of one workload on the GPU box - the timeline of one step as benchmarked
# (concurrent streams) and the per-kernel durations with the streams serialised.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r3; mkdir -p $O
A="--no-e2e --no-cpu-baseline --no-clock-probe --steps ${PROF_NSTEPS:-4} --warmup 2"
rm -rf $O/tn $O/ts
rocprofv3 --kernel-trace --output-format csv -d $O/tn -- python3 bench.py $A "$@" > $O/tn.json 2> $O/tn.err
python3 tools/timeline.py $(find $O/tn -name "*kernel_trace.csv" | head -1) ${PROF_BACK:-2} ${PROF_STEPS:-1} | tee $O/timeline_tn.txt
export PLAAC_SERIAL_STREAMS=1
rocprofv3 --kernel-trace --output-format csv -d $O/ts -- python3 bench.py $A "$@" > $O/ts.json 2> $O/ts.err
echo "---- serialised streams"
python3 tools/timeline.py $(find $O/ts -name "*kernel_trace.csv" | head -1) 2 | tee $O/timeline_ts.txt
rm -rf $O/tn $O/ts
