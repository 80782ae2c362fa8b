#!/bin/bash
# round 6 evidence at the final tree: counter passes + kernel statistics for the headline (cfg4, all 10 M sequences), track mode,
# the 9-point sweep over the 1.25 M share and config 3 (default forms), the SQ counters of the track kernels, then the bench
# lines of every configuration.        bash tools/r06_evidence.sh pmc4 | pmcrest | sq | lines
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r6; mkdir -p $O
case "$1" in
pmc4)
  PMC_DIR=$O/pmc_cfg4 bash tools/pmc.sh > $O/pmc_cfg4.txt 2>&1; tail -n 5 $O/pmc_cfg4.txt ;;
pmcrest)
  PMC_DEFAULT_FORMS=1 PMC_DIR=$O/pmc_tracks bash tools/pmc.sh --tracks --nprot 1250000 > $O/pmc_tracks.txt 2>&1; tail -n 3 $O/pmc_tracks.txt
  PMC_DEFAULT_FORMS=1 PMC_DIR=$O/pmc_sweep bash tools/pmc.sh --sweep --nprot 1250000 > $O/pmc_sweep.txt 2>&1; tail -n 3 $O/pmc_sweep.txt
  PMC_DEFAULT_FORMS=1 PMC_DIR=$O/pmc_cfg3 bash tools/pmc.sh --config 3 > $O/pmc_cfg3.txt 2>&1; tail -n 3 $O/pmc_cfg3.txt
  PMC_DEFAULT_FORMS=1 PMC_DIR=$O/pmc_share bash tools/pmc.sh --nprot 1250000 > $O/pmc_share.txt 2>&1; tail -n 3 $O/pmc_share.txt ;;
sq)
  bash tools/r06_pmc_tracks_sq.sh > /dev/null 2>&1; head -c 3000 $O/pmc_tracks_sq.txt ;;
lines)
  Q="--no-e2e --no-predict --no-tracks-leg --no-tolerance-leg"
  python3 bench.py > $O/bench_cfg4_full.json 2> $O/bench_cfg4_full.err; echo "cfg4 rc=$?"
  python3 bench.py --config 2 $Q > $O/bench_cfg2.json 2>> $O/lines.err; echo "cfg2 rc=$?"
  python3 bench.py --config 3 $Q > $O/bench_cfg3_two_pass.json 2>> $O/lines.err; echo "cfg3 rc=$?"
  python3 bench.py --nprot 1250000 $Q > $O/bench_cfg4_share_1250k.json 2>> $O/lines.err; echo "share rc=$?"
  python3 bench.py --tracks --nprot 1250000 $Q > $O/bench_tracks_1250k.json 2>> $O/lines.err; echo "tracks rc=$?"
  python3 bench.py --tracks --nprot 1250000 --max-len 8192 $Q > $O/bench_tracks_1250k_clipped_8192.json 2>> $O/lines.err; echo "tracks clipped rc=$?"
  python3 bench.py --sweep --nprot 1250000 $Q > $O/bench_sweep_1250k.json 2>> $O/lines.err; echo "sweep share rc=$?"
  python3 bench.py --sweep $Q --steps 5 > $O/bench_sweep_10M.json 2>> $O/lines.err; echo "sweep 10M rc=$?"
  python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r6/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        r=d['roofline']
        print(f.split('/')[-1], d['ms_per_step'], 'frac', r['frac'], 'traffic_all', r.get('traffic_all_kernels'), 'issue', (r.get('issue') or {}).get('frac'), 'match', (d.get('cpu_baseline') or {}).get('gpu_rows_match_oracle'), 'tracks', (d.get('tracks') or {}).get('frac'), 'stale', r.get('pmc_counters_stale'))
    except Exception as e: print(f, 'ERR', e)
PY
  ;;
ranks)
  # plumbing only: six ranks share one device over gloo (the GPU box allows six processes on its card; eight would be killed)
  python3 bench.py --gpus 6 --backend gloo --one-device --nprot 200000 --steps 5 --warmup 1 --no-e2e --no-predict --no-tracks-leg --no-host-leg --no-clock-probe > $O/bench_6rank_one_device.json 2> $O/bench_6rank.err; echo "6 ranks rc=$?"; tail -c 600 $O/bench_6rank_one_device.json
  # ... and ONE proteome cut over the six ranks (strong scaling; from four ranks on the rows end in ranges, by an all-to-all)
  python3 bench.py --gpus 6 --backend gloo --one-device --config 2 --steps 5 --warmup 1 --no-e2e > $O/bench_6rank_one_device_strong_ranges.json 2> $O/bench_6rank_strong.err; echo "6 ranks strong rc=$?"; tail -c 900 $O/bench_6rank_one_device_strong_ranges.json ;;
esac
