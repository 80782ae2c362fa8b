cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r3; mkdir -p $O; rm -rf $O/q
timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "not lane" 2>&1 | tail -3
PLAAC_SERIAL_STREAMS=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/q -- python3 bench.py --no-e2e --steps 3 --warmup 1 --no-cpu-baseline --tracks > /dev/null 2>&1
python3 - <<'PY'
import csv,glob
for f in glob.glob('gpurun_out/r3/q/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'anonymous' in r['Name'] and 'at::' not in r['Name'] and float(r['TotalDurationNs'])>3e5:
            print(r['Name'][:70].ljust(70), r['Calls'], round(float(r['AverageNs'])/1e6,3))
PY
rm -rf $O/q
timeout 300 python3 bench.py --no-e2e --steps 10 --tracks 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('conc', d['ms_per_step'], d['roofline']['kernel_ms'], (d.get('cpu_baseline') or {}).get('gpu_rows_match_oracle'))"
