#!/bin/bash
# round 5, first GPU call: the GPU suite at the new tree, then the default bench line (tracks leg, verified step)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r5; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $O/gputests.txt 2>&1; echo "pytest rc=$?" >> $O/gputests.txt
tail -n 5 $O/gputests.txt
timeout -k 10 600 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r5/bench_default.json').read().strip().splitlines()[-1])
print('ms_per_step', d['ms_per_step'], 'match', d['cpu_baseline']['gpu_rows_match_oracle'], 'verified', d['config']['verified_step'])
print('tracks', {k:v for k,v in (d.get('tracks') or {}).items() if k not in ('what','workload','tolerance')})
print('predicted', d.get('predicted_strong_efficiency'), 'e2e', (d.get('e2e') or {}).get('wall_s'))
PY
