#!/bin/bash
# tools/kb_probe.sh - cost breakdown of the window kernel (K-B) on the bench workload, run on the GPU box.
# Uses the probe build (make probe): PLAAC_KB_SKIP leaves parts of the kernel out; the drop in the serialised
# kernel time is that part's cost. Results of these runs are wrong on purpose and never checked.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export PLAAC_NATIVE_LIB=$GRAFT_REPO_ROOT/build/libplaac_native_probe.so PLAAC_SERIAL_STREAMS=1
for m in 0 1 2 3 4 8 16 32 64 127; do
  PLAAC_KB_SKIP=$m python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('skip=%3d  k_tracks %.3f ms' % ($m, d['roofline']['kernel_ms']['tracks']))"
done | tee gpurun_out/kb_probe.txt
