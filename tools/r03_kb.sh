#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r3; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -5
run() { # label, env..., -- bench args
  timeout -k 10 300 env "${@:2}" > $O/kb.json 2> $O/kb.err; python3 -c "
import json;d=json.loads(open('$O/kb.json').read().strip().splitlines()[-1]);print('$1', d['ms_per_step'], d['roofline']['kernel_ms'], d['config'].get('exact_tier_fallbacks_rank0'), (d.get('cpu_baseline') or {}).get('gpu_rows_match_oracle'))"
}
B="python3 bench.py --no-e2e --steps 20"
run "10M lane prio" $B
run "10M lane noprio" PLAAC_KB_PRIORITY=0 $B --no-cpu-baseline
run "1250k lane" $B --nprot 1250000 --no-cpu-baseline
PLAAC_SERIAL_STREAMS=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/q -- python3 bench.py --no-e2e --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python3 - <<'PY'
import csv,glob
for f in glob.glob('gpurun_out/r3/q/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'anonymous' in r['Name'] and 'at::' not in r['Name'] and float(r['TotalDurationNs'])>3e5:
            print(r['Name'][:70].ljust(70), r['Calls'], round(float(r['AverageNs'])/1e6,3))
PY
rm -rf $O/q
