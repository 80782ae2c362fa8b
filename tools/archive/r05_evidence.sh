#!/bin/bash
# round 5 evidence at the final tree: counter passes + kernel statistics for the headline (cfg4, all 10 M sequences), track mode,
# the 9-point sweep over the 1.25 M share and config 3 (default forms), then the bench lines of every configuration.
#   bash tools/r05_evidence.sh pmc4 | pmcrest | lines
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r5; mkdir -p $O
case "$1" in
pmc4)
  PMC_DIR=$O/pmc_cfg4 bash tools/pmc.sh > $O/pmc_cfg4.txt 2>&1; tail -n 5 $O/pmc_cfg4.txt ;;
pmcrest)
  PMC_DEFAULT_FORMS=1 PMC_DIR=$O/pmc_tracks bash tools/pmc.sh --tracks --nprot 1250000 > $O/pmc_tracks.txt 2>&1; tail -n 3 $O/pmc_tracks.txt
  PMC_DEFAULT_FORMS=1 PMC_DIR=$O/pmc_sweep bash tools/pmc.sh --sweep --nprot 1250000 > $O/pmc_sweep.txt 2>&1; tail -n 3 $O/pmc_sweep.txt
  PMC_DEFAULT_FORMS=1 PMC_DIR=$O/pmc_cfg3 bash tools/pmc.sh --config 3 > $O/pmc_cfg3.txt 2>&1; tail -n 3 $O/pmc_cfg3.txt
  PMC_DEFAULT_FORMS=1 PMC_DIR=$O/pmc_share bash tools/pmc.sh --nprot 1250000 > $O/pmc_share.txt 2>&1; tail -n 3 $O/pmc_share.txt ;;
lines)
  Q="--no-e2e --no-predict --no-tracks-leg"
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
for f in sorted(glob.glob('gpurun_out/r5/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        r=d['roofline']
        print(f.split('/')[-1], d['ms_per_step'], 'frac', r['frac'], 'traffic_all', r.get('traffic_all_kernels'), 'issue', (r.get('issue') or {}).get('frac'), 'match', (d.get('cpu_baseline') or {}).get('gpu_rows_match_oracle'), 'tracks', (d.get('tracks') or {}).get('frac'))
    except Exception as e: print(f, 'ERR', e)
PY
  ;;
esac
