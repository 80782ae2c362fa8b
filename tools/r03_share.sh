#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r3; mkdir -p $O
export PLAAC_STREAM_DEBUG=1
run() { # label, env..., -- bench args
  timeout -k 10 300 env "${@:2}" > $O/share.json 2> $O/share.err; grep "^plaac: " $O/share.err | head -4; python3 -c "
import json;d=json.loads(open('$O/share.json').read().strip().splitlines()[-1]);print('$1', d['ms_per_step'], d['roofline']['kernel_ms'])"
}
B="python3 bench.py --no-e2e --no-cpu-baseline --steps 20"
run "10M" $B
run "10M again" $B
(cd _ab/old; run "10M old" $B)
run "1250k" $B --nprot 1250000 --contexts 1
run "1250k ctx2" $B --nprot 1250000 --contexts 2
(cd _ab/old; run "1250k old" $B --nprot 1250000)
run "cfg3" $B --config 3 --steps 100 --contexts 1
run "cfg2" $B --config 2 --steps 200 --contexts 1
run "tracks" $B --tracks --steps 10
run "sweep1250k" $B --sweep --nprot 1250000 --steps 5
timeout -k 10 600 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -3
