#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r3; mkdir -p $O
timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "histogram" 2>&1 | tail -3
for e in "0 4" "1 4" "2 4" "0 8"; do set -- $e
PLAAC_HIST_EXP=$1 PLAAC_HIST_BPC=$2 timeout -k 10 300 python3 bench.py --no-e2e --no-cpu-baseline --steps 2 --warmup 1 > $O/h.json 2>/dev/null; python3 -c "
import json;d=json.loads(open('$O/h.json').read().strip().splitlines()[-1]);print('hist exp=$1 bpc=$2', d['roofline']['histogram_pass']['ms'], d['roofline']['histogram_pass']['achieved_GBps'])"
done
