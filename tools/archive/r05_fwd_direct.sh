#!/bin/bash
# round 5 experiment: the forward pass straight from the batch (64 contiguous bytes per lane and load group) against the packed
# copy: parity, the kernel by itself (serialised streams, rocprofv3 kernel stats), the step
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r5; mkdir -p $O
PLAAC_FWD_DIRECT=1 PLAAC_LATENCY_MODE=0 timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "edge_lengths or random_batches or adversarial or overlapp" > $O/fd_tests.txt 2>&1; tail -n 3 $O/fd_tests.txt
out=$O/fwd_direct.txt; : > $out
F="--no-e2e --no-cpu-baseline --no-clock-probe --no-host-leg --no-predict --no-tracks-leg"
for v in 0 1; do
  rm -rf $O/fd
  PLAAC_FWD_DIRECT=$v PLAAC_SERIAL_STREAMS=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/fd -- python3 bench.py $F --steps 3 --warmup 1 > /dev/null 2>$O/fd.err
  echo "== PLAAC_FWD_DIRECT=$v, serialised streams: kernel averages (ms)" >> $out
  python3 - >> $out <<'PY'
import csv,glob
for f in glob.glob('gpurun_out/r5/fd/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'anonymous' in r['Name'] and 'at::' not in r['Name'] and float(r['TotalDurationNs'])>3e5:
            print('  ', r['Name'][:70].ljust(70), r['Calls'], round(float(r['AverageNs'])/1e6,3))
PY
done
rm -rf $O/fd
for rep in 1 2 3; do for v in 0 1; do
  PLAAC_FWD_DIRECT=$v timeout -k 10 300 python3 bench.py $F --steps 16 --warmup 4 2>>$O/fd.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('PLAAC_FWD_DIRECT=$v:', 'ms/step', d['ms_per_step'], d['roofline']['kernel_ms'])" >> $out
done; done
cat $out
