#!/bin/bash
# tools/r02_evidence.sh - round-2 evidence on the GPU box: bench lines of every BASELINE config, rocprofv3 kernel stats
# and PMC passes for the headline workload and for track mode, the 2-rank (one device, gloo) plumbing run.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r2/ev; rm -rf $O; mkdir -p $O
# PMC passes first: the bench lines quote their counters (profiles/*pmc_traffic*.json)
# rocprofv3: headline workload (10 M sequences)
bash tools/pmc.sh > $O/pmc_cfg4_full.log 2>&1
mkdir -p $O/pmc_cfg4_full; cp gpurun_out/pmc/summary.json gpurun_out/pmc/pmc_traffic.json $O/pmc_cfg4_full/ 2>/dev/null
cp gpurun_out/pmc/pmc_traffic.json profiles/pmc_traffic.json   # the bench lines below quote THESE counters
cp $(find gpurun_out/pmc/trace_concurrent -name "*kernel_stats.csv" | head -1) $O/pmc_cfg4_full/kernel_stats_concurrent.csv
cp $(find gpurun_out/pmc/trace_serial -name "*kernel_stats.csv" | head -1) $O/pmc_cfg4_full/kernel_stats_serial.csv
# rocprofv3: track mode at the 1.25 M share
bash tools/pmc.sh --tracks --steps 2 > $O/pmc_tracks.log 2>&1
mkdir -p $O/pmc_tracks; cp gpurun_out/pmc/summary.json gpurun_out/pmc/pmc_traffic.json $O/pmc_tracks/ 2>/dev/null
cp gpurun_out/pmc/pmc_traffic.json profiles/r02_pmc_traffic_tracks_1250k.json
cp $(find gpurun_out/pmc/trace_serial -name "*kernel_stats.csv" | head -1) $O/pmc_tracks/kernel_stats_serial.csv
rm -rf gpurun_out/pmc
python3 bench.py > $O/bench_cfg4_full.json 2> $O/bench_cfg4_full.err
python3 bench.py --config 2 --steps 200 --no-e2e > $O/bench_cfg2.json 2>/dev/null
python3 bench.py --config 3 --steps 100 --no-e2e > $O/bench_cfg3_two_pass.json 2>/dev/null
python3 bench.py --nprot 1250000 --no-e2e > $O/bench_cfg4_shard_1250k.json 2>/dev/null
python3 bench.py --tracks --steps 10 --no-e2e > $O/bench_tracks_1250k.json 2>/dev/null
python3 bench.py --sweep --nprot 1250000 --steps 5 --no-e2e > $O/bench_sweep_1250k.json 2>/dev/null
python3 bench.py --sweep --steps 5 --no-e2e > $O/bench_sweep_10M.json 2>/dev/null
python3 bench.py --sweep --naive-sweep --nprot 1250000 --steps 3 --no-e2e > $O/bench_sweep_naive_1250k.json 2>/dev/null
# two ranks on the one device, exchange over gloo (RCCL needs one device per rank): HIP contexts + row gather together
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py \
  --gpus 2 --one-device --backend gloo --nprot 300000 --steps 5 --no-e2e --no-cpu-baseline > $O/bench_2rank_one_device_gloo.json 2> $O/bench_2rank.err
for f in $O/*.json; do echo "== $f"; python3 -c "
import json,sys
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], (d.get('cpu_baseline') or {}).get('gpu_rows_match_oracle'))
except Exception as e: print('ERR', e)
"; done
tail -3 $O/bench_2rank.err
