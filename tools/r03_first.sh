#!/bin/bash
# tools/r03_first.sh - first GPU pass of round 3: parity tests, issue-cost probe, bench lines (default, 1.25 M share with one
# and two contexts), the 2-rank strong-scaling plumbing on one device, serial kernel stats
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r3; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -5 $O/tests.log
timeout -k 10 300 build/issue_probe > $O/issue_probe.txt 2>&1; echo "probe rc=$?"
timeout -k 10 600 python3 bench.py --steps 10 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
timeout -k 10 300 python3 bench.py --nprot 1250000 --no-e2e --contexts 1 > $O/bench_1250k_ctx1.json 2>/dev/null; echo "rc=$?"
timeout -k 10 300 python3 bench.py --nprot 1250000 --no-e2e --contexts 2 > $O/bench_1250k_ctx2.json 2>/dev/null; echo "rc=$?"
timeout -k 10 300 python3 bench.py --nprot 1250000 --no-e2e --contexts 3 > $O/bench_1250k_ctx3.json 2>/dev/null; echo "rc=$?"
timeout -k 10 300 python3 bench.py --config 3 --steps 100 --no-e2e > $O/bench_cfg3.json 2>/dev/null; echo "rc=$?"
timeout -k 10 300 python3 bench.py --config 2 --steps 200 --no-e2e > $O/bench_cfg2.json 2>/dev/null; echo "rc=$?"
timeout -k 10 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py \
  --gpus 2 --one-device --backend gloo --steps 5 --no-e2e > $O/bench_2rank_strong.json 2> $O/bench_2rank_strong.err; echo "2rank rc=$?"
PLAAC_SERIAL_STREAMS=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/q -- python3 bench.py --no-e2e --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python3 - <<'PY'
import csv,glob,json
for f in glob.glob('gpurun_out/r3/q/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if float(r['TotalDurationNs'])>1e5:
            print(r['Name'][:80].ljust(80), r['Calls'], round(float(r['AverageNs'])/1e6,3))
for f in sorted(glob.glob('gpurun_out/r3/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline'].get('histogram_pass'), (d.get('cpu_baseline') or {}).get('gpu_rows_match_oracle'), (d.get('cpu_baseline') or {}).get('max_residue_offset_checked'), d.get('weak'))
    except Exception as e: print(f, 'ERR', e)
PY
tail -5 $O/bench_2rank_strong.err
