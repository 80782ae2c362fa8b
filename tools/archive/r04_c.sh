#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
PLAAC_MIXED=0 bash tools/r04_trace.sh cfg3_unmixed 2 --config 3
bash tools/r04_trace.sh cfg3_mixed 2 --config 3
grep -h "k_fwd_pair\|k_long\|k_vit<1, true\|k_win<1, [23]>" gpurun_out/r4/timeline_cfg3_unmixed.txt gpurun_out/r4/timeline_cfg3_mixed.txt
