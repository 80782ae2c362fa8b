cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r6
run() { L=$1; shift; timeout -k 10 150 python3 bench.py --backend gloo --one-device --steps 2 --warmup 1 --no-e2e --no-weak-leg "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['cpu_baseline']; print('$L', d['n_gpus'], 'match', c['gpu_rows_match_oracle'], c['mismatching_slices'], d['config']['exchange'][:30])"; }
run c3_4_rank0 --gpus 4 --config 3 --exchange rank0
run c3_2_ranges --gpus 2 --config 3 --exchange ranges
run c3_2_rank0 --gpus 2 --config 3 --exchange rank0
run c2_4_ranges --gpus 4 --config 2 --exchange ranges
run c3_4_ranges --gpus 4 --config 3 --exchange ranges
