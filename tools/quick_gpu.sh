#!/bin/bash
# tools/quick_gpu.sh - development loop on the GPU box: parity tests, per-kernel durations (serialised streams) and the bench line
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2; rm -rf gpurun_out/r2/q
timeout 1200 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -5
PLAAC_SERIAL_STREAMS=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2/q -- python3 bench.py --no-e2e --steps 3 --warmup 1 --no-cpu-baseline "$@" > /dev/null 2>&1
python3 - <<'PY'
import csv,glob
for f in glob.glob('gpurun_out/r2/q/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'anonymous' in r['Name'] and float(r['TotalDurationNs'])>1e5:
            print(r['Name'][:70].ljust(70), r['Calls'], round(float(r['AverageNs'])/1e6,3))
PY
timeout 300 python3 bench.py --no-e2e --steps 10 "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('conc', d['ms_per_step'], d['roofline']['kernel_ms'], (d.get('cpu_baseline') or {}).get('gpu_rows_match_oracle'))"
