#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r4
timeout -k 10 600 python3 -m pytest tests -x -q -m gpu --timeout=200 > $O/c_all.txt 2>&1; tail -6 $O/c_all.txt
timeout -k 10 800 python3 tools/r04_e2e_ab.py 3000000 2>&1 | tee $O/c_e2e.txt
