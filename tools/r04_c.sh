#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r4
timeout -k 10 600 python3 -m pytest tests -x -q -m gpu --timeout=200 > $O/c_all.txt 2>&1; tail -6 $O/c_all.txt
PLAAC_MIXED_GROUPS=3 PLAAC_MIXED_MIN_REST=1 timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_real_proteomes.py -x -q -m gpu --timeout=200 > $O/c_mixed.txt 2>&1; tail -4 $O/c_mixed.txt
PLAAC_OVERLAP=1 timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu --timeout=200 > $O/c_ov.txt 2>&1; tail -4 $O/c_ov.txt
