#!/bin/bash
# the whole -m gpu suite on the GPU box, output under gpurun_out/ (tools/gpu_suite.sh [pytest args])
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r6; mkdir -p $O
t0=$(date +%s)
timeout -k 10 1000 python3 -m pytest tests -x -q -m gpu --durations=15 "$@" > $O/gpu_suite.txt 2>&1
rc=$?
echo "exit $rc after $(( $(date +%s) - t0 )) s" >> $O/gpu_suite.txt
tail -n 30 $O/gpu_suite.txt
exit $rc
