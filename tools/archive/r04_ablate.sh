#!/bin/bash
# What does each kernel ADD to a 10 M-sequence step beside the others? (PLAAC_DEBUG_SKIP: the kernel is not launched from the
# third call on; rows are stale.) One box, one process per variant.
mkdir -p gpurun_out/r4
out=gpurun_out/r4/ablate.txt
: > $out
F="--steps 12 --warmup 4 --no-cpu-baseline --no-e2e --no-predict --no-clock-probe --no-host-leg"
for skip in "" k_win k_fwd k_vit k_core_list k_vit,k_core_list,k_finish k_refine_centres k_tracksL,k_refine_centres k_tracksL,k_refine_centres,k_tracks20f k_pack k_win,k_fwd,k_vit,k_core_list,k_finish ""; do
  echo "== skip: [$skip]" >> $out
  PLAAC_NATIVE_LIB=$PWD/plaac_amd/libplaac_native_diag.so PLAAC_DEBUG_SKIP="$skip" timeout -k 10 300 python3 bench.py --allow-diagnostics $F 2>>gpurun_out/r4/ablate.err | python3 -c "
import sys,json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print('   ms_per_step', d['ms_per_step'], 'issue', (d['roofline'].get('issue') or {}).get('frac'))
" >> $out || echo "   failed" >> $out
  echo "done [$skip]"
done
cat $out
