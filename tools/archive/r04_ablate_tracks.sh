#!/bin/bash
# What does each kernel ADD to a track-mode step over the 1.25 M-sequence share? (PLAAC_DEBUG_SKIP, results stale)
mkdir -p gpurun_out/r4
out=gpurun_out/r4/ablate_tracks.txt
: > $out
F="--steps 10 --warmup 3 --no-cpu-baseline --no-e2e --no-predict --no-clock-probe --no-host-leg --tracks --nprot 1250000"
for skip in "" k_tracks20s k_post k_fwd_pair,k_fwd k_bwd_pair,k_bwd k_vit k_win k_pack k_finish k_core_par,k_core_chain,k_core_eval,k_core_reduce k_tracks20_whole ""; do
  echo "== skip: [$skip]" >> $out
  PLAAC_NATIVE_LIB=$PWD/plaac_amd/libplaac_native_diag.so PLAAC_DEBUG_SKIP="$skip" timeout -k 10 300 python3 bench.py --allow-diagnostics $F 2>>gpurun_out/r4/ablate_tracks.err | python3 -c "
import sys,json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print('   ms_per_step', d['ms_per_step'])
" >> $out || echo "   failed" >> $out
done
cat $out
