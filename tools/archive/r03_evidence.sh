#!/bin/bash
# tools/r03_evidence.sh - round-3 evidence on the GPU box: PMC passes + kernel stats of the headline workload, bench lines
# of every BASELINE config, probes, the 2-rank strong-scaling plumbing run, the low-complexity fallback check.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r3/ev; rm -rf $O; mkdir -p $O
bash tools/pmc.sh > $O/pmc_cfg4_full.log 2>&1
cp gpurun_out/pmc/summary.json $O/pmc_summary_cfg4_full.json; cp gpurun_out/pmc/pmc_traffic.json $O/pmc_traffic.json
cp gpurun_out/pmc/pmc_traffic.json profiles/pmc_traffic.json   # the bench lines below quote THESE counters
cp $(find gpurun_out/pmc/trace_concurrent -name "*kernel_stats.csv" | head -1) $O/kernel_stats_concurrent_cfg4_full.csv
cp $(find gpurun_out/pmc/trace_serial -name "*kernel_stats.csv" | head -1) $O/kernel_stats_serial_cfg4_full.csv
python3 tools/timeline.py $(find gpurun_out/pmc/trace_concurrent -name "*kernel_trace.csv" | head -1) 3 2 > $O/timeline_concurrent_cfg4_full.txt
rm -rf gpurun_out/pmc
# track mode (1.25 M-sequence share): the same passes
bash tools/pmc.sh --tracks > $O/pmc_tracks_1250k.log 2>&1
cp gpurun_out/pmc/summary.json $O/pmc_summary_tracks_1250k.json; cp gpurun_out/pmc/pmc_traffic.json $O/pmc_traffic_tracks_1250k.json
cp gpurun_out/pmc/pmc_traffic.json profiles/r03_pmc_traffic_tracks_1250k.json
cp $(find gpurun_out/pmc/trace_serial -name "*kernel_stats.csv" | head -1) $O/kernel_stats_serial_tracks_1250k.csv
python3 tools/timeline.py $(find gpurun_out/pmc/trace_concurrent -name "*kernel_trace.csv" | head -1) 2 > $O/timeline_concurrent_tracks_1250k.txt
rm -rf gpurun_out/pmc
python3 bench.py > $O/bench_cfg4_full.json 2> $O/bench_cfg4_full.err
python3 bench.py --no-overlap --no-e2e --no-cpu-baseline > $O/bench_cfg4_full_no_overlap.json 2>/dev/null
python3 bench.py --config 2 --steps 200 --no-e2e > $O/bench_cfg2.json 2>/dev/null
python3 bench.py --config 3 --steps 100 --no-e2e > $O/bench_cfg3_two_pass.json 2>/dev/null
python3 bench.py --nprot 1250000 --no-e2e > $O/bench_cfg4_shard_1250k.json 2>/dev/null
python3 bench.py --tracks --steps 10 --no-e2e > $O/bench_tracks_1250k.json 2>/dev/null
python3 bench.py --sweep --nprot 1250000 --steps 5 --no-e2e > $O/bench_sweep_1250k.json 2>/dev/null
python3 bench.py --sweep --steps 5 --no-e2e > $O/bench_sweep_10M.json 2>/dev/null
PLAAC_KB_LANE=0 python3 bench.py --no-e2e --no-cpu-baseline > $O/bench_cfg4_full_stream_form.json 2>/dev/null
# two ranks on the one device, exchange over gloo (RCCL needs one device per rank): ONE proteome cut over the ranks
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py \
  --gpus 2 --one-device --backend gloo --steps 5 --no-e2e > $O/bench_2rank_strong_one_device_gloo.json 2> $O/bench_2rank.err
build/issue_probe > $O/issue_probe.txt 2>&1
build/queue_probe > $O/queue_probe.txt 2>&1
python3 tools/check_filter_large.py > $O/check_filter_large.txt 2>&1
python3 -m pytest tests/test_real_proteomes.py -q -m gpu -s 2>&1 | grep "exact tier" > $O/real_proteome_fallbacks.txt
for f in $O/bench_*.json; do echo "== $f"; python3 -c "
import json,sys
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline'].get('histogram_pass',{}).get('ms'), (d['roofline'].get('issue') or {}).get('frac'), (d.get('cpu_baseline') or {}).get('gpu_rows_match_oracle'), (d.get('e2e') or {}).get('value'))
except Exception as e: print('ERR', e)
"; done
tail -3 $O/bench_2rank.err
