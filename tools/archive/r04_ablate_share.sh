#!/bin/bash
# What does each kernel ADD to a step over the 1.25 M-sequence share (mixed forms, overlapping steps)? (PLAAC_DEBUG_SKIP, results stale)
mkdir -p gpurun_out/r4
out=gpurun_out/r4/ablate_share.txt
: > $out
F="--steps 40 --warmup 6 --no-cpu-baseline --no-e2e --no-predict --no-clock-probe --no-host-leg --nprot 1250000"
for skip in "" k_long k_vit k_core_par,k_core_chain,k_core_eval,k_core_reduce k_finish k_fwd k_win k_core_list k_tracksL k_tracks20f k_refine_centres k_tracks20_list k_pack k_plan_lengths,k_plan_scatter ""; do
  echo "== skip: [$skip]" >> $out
  PLAAC_NATIVE_LIB=$PWD/plaac_amd/libplaac_native_diag.so PLAAC_DEBUG_SKIP="$skip" timeout -k 10 300 python3 bench.py --allow-diagnostics $F 2>>gpurun_out/r4/ablate_share.err | python3 -c "
import sys,json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print('   ms_per_step', d['ms_per_step'])
" >> $out || echo "   failed" >> $out
done
cat $out
