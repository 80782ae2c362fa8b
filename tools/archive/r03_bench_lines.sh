#!/bin/bash
# tools/r03_bench_lines.sh - the bench lines of every configuration at HEAD (no PMC passes): gpurun_out/r3/ev/bench_*.json
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r3/ev; mkdir -p $O
python3 bench.py > $O/bench_cfg4_full.json 2> $O/bench_cfg4_full.err
python3 bench.py --no-overlap --no-e2e --no-cpu-baseline > $O/bench_cfg4_full_no_overlap.json 2>/dev/null
python3 bench.py --config 2 --steps 200 --no-e2e > $O/bench_cfg2.json 2>/dev/null
python3 bench.py --config 3 --steps 100 --no-e2e > $O/bench_cfg3_two_pass.json 2>/dev/null
python3 bench.py --nprot 1250000 --no-e2e > $O/bench_cfg4_shard_1250k.json 2>/dev/null
python3 bench.py --tracks --steps 10 --no-e2e > $O/bench_tracks_1250k.json 2>/dev/null
PLAAC_KB_LANE=0 python3 bench.py --no-e2e --no-cpu-baseline > $O/bench_cfg4_full_stream_form.json 2>/dev/null
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py \
  --gpus 2 --one-device --backend gloo --steps 5 --no-e2e > $O/bench_2rank_strong_one_device_gloo.json 2> $O/bench_2rank.err
for f in $O/bench_*.json; do echo "== $f"; python3 -c "
import json,sys
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], (d['roofline'].get('issue') or {}).get('frac'), (d['roofline'].get('issue') or {}).get('frac_at_measured_clock'), (d.get('cpu_baseline') or {}).get('gpu_rows_match_oracle'), (d.get('e2e') or {}).get('value'))
except Exception as e: print('ERR', e)
"; done
