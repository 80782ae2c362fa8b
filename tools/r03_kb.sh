#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r3; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -3
run() { # label, env..., -- bench args
  timeout -k 10 300 env "${@:2}" > $O/kb.json 2> $O/kb.err; python3 -c "
import json;d=json.loads(open('$O/kb.json').read().strip().splitlines()[-1]);print('$1', d['ms_per_step'], d['roofline']['kernel_ms'], d['config'].get('exact_tier_fallbacks_rank0'), (d.get('cpu_baseline') or {}).get('gpu_rows_match_oracle'))"
}
B="python3 bench.py --no-e2e --steps 20"
run "1250k" $B --nprot 1250000
run "1250k win2" PLAAC_WIN_THREE=0 $B --nprot 1250000 --no-cpu-baseline
run "cfg3" $B --config 3 --steps 100
run "cfg3 win2" PLAAC_WIN_THREE=0 $B --config 3 --steps 100 --no-cpu-baseline
run "cfg2" $B --config 2 --steps 200
run "10M" $B
