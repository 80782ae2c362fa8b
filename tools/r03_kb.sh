#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r3; mkdir -p $O
run() { # label, env..., -- bench args
  timeout -k 10 300 env "${@:2}" > $O/kb.json 2> $O/kb.err; python3 -c "
import json;d=json.loads(open('$O/kb.json').read().strip().splitlines()[-1]);print('$1', d['ms_per_step'], d['roofline']['kernel_ms'], d['config'].get('exact_tier_fallbacks_rank0'), (d.get('cpu_baseline') or {}).get('gpu_rows_match_oracle'))"
}
B="python3 bench.py --no-e2e --steps 20 --no-cpu-baseline"
run "10M ctx1" $B
run "10M ctx2" $B --contexts 2
run "10M ctx3" $B --contexts 3
run "1250k ctx2" $B --nprot 1250000 --contexts 2
