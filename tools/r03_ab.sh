#!/bin/bash
# A/B on one box: round-2 HEAD (_ab/old, built from `git archive 228aa24`) against the working tree, same bench flags
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r3; mkdir -p $O
export PLAAC_STREAM_DEBUG=1
for i in 1 2; do
  if [ -d _ab/old ]; then (cd _ab/old && timeout -k 10 300 python3 bench.py --no-e2e --no-cpu-baseline --steps 20 > $O/ab_old_$i.json 2>/dev/null; python3 -c "
import json;d=json.loads(open('$O/ab_old_$i.json').read().strip().splitlines()[-1]);print('old', d['ms_per_step'], d['roofline']['kernel_ms'])"); fi
  timeout -k 10 300 python3 bench.py --no-e2e --no-cpu-baseline --steps 20 "$@" > $O/ab_new_$i.json 2> $O/ab_new_$i.err; grep "^plaac:" $O/ab_new_$i.err; python3 -c "
import json;d=json.loads(open('$O/ab_new_$i.json').read().strip().splitlines()[-1]);print('new', d['ms_per_step'], d['roofline']['kernel_ms'])"
done
