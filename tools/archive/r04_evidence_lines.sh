#!/bin/bash
# tools/r04_evidence_lines.sh - round-4 evidence, part 2 (one gpurun call): the bench line of every BASELINE configuration at
# HEAD (profiles/pmc_traffic*.json of part 1 must be in place: the lines quote those counters), probes, stress, e2e A/B.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r4/ev; mkdir -p $O
echo "default line"; PLAAC_TIMING=1 python3 bench.py > $O/bench_cfg4_full.json 2> $O/bench_cfg4_full.err
echo "no overlap"; python3 bench.py --no-overlap --no-e2e --no-cpu-baseline --no-predict > $O/bench_cfg4_full_no_overlap.json 2>/dev/null
echo "cfg2"; python3 bench.py --config 2 --steps 200 --no-e2e > $O/bench_cfg2.json 2>/dev/null
echo "cfg3"; python3 bench.py --config 3 --steps 100 --no-e2e > $O/bench_cfg3_two_pass.json 2>/dev/null
echo "share"; python3 bench.py --nprot 1250000 --steps 40 --no-e2e > $O/bench_cfg4_shard_1250k.json 2>/dev/null
echo "share, round-3 forms"; PLAAC_MIXED=0 python3 bench.py --nprot 1250000 --steps 40 --no-e2e --no-cpu-baseline > $O/bench_cfg4_shard_1250k_unmixed.json 2>/dev/null
echo "tracks"; python3 bench.py --tracks --steps 10 --no-e2e > $O/bench_tracks_1250k.json 2>/dev/null
echo "sweep share"; python3 bench.py --sweep --nprot 1250000 --steps 5 --no-e2e > $O/bench_sweep_1250k.json 2>/dev/null
echo "sweep 10M"; python3 bench.py --sweep --steps 5 --no-e2e > $O/bench_sweep_10M.json 2>/dev/null
echo "2 ranks"; python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py \
  --gpus 2 --one-device --backend gloo --steps 5 --no-e2e > $O/bench_2rank_strong_one_device_gloo.json 2> $O/bench_2rank.err
echo "probes"; build/queue_probe > $O/queue_probe.txt 2>&1
echo "stress"; PLAAC_MIXED_MIN_REST=8 timeout 600 python3 tools/stress_overlap.py 40 4 > $O/stress_overlap_mixed.txt 2>&1; tail -2 $O/stress_overlap_mixed.txt
echo "filter large"; timeout 600 python3 tools/check_filter_large.py > $O/check_filter_large.txt 2>&1
echo "e2e A/B"; timeout 600 python3 tools/r04_e2e_ab.py 3000000 > $O/e2e_pipeline_ab.txt 2>&1
timeout 300 python3 -m pytest tests/test_real_proteomes.py -q -m gpu -s 2>&1 | grep "exact tier" > $O/real_proteome_fallbacks.txt
for f in $O/bench_*.json; do echo "== $f"; python3 -c "
import json,sys
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], (d['roofline'].get('issue') or {}).get('frac'), (d['roofline'].get('issue') or {}).get('frac_at_measured_clock'), (d.get('cpu_baseline') or {}).get('gpu_rows_match_oracle'), (d.get('e2e') or {}).get('value'), d.get('predicted_strong_efficiency'))
except Exception as e: print('ERR', e)
"; done
tail -3 $O/bench_2rank.err
