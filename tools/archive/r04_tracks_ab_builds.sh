#!/bin/bash
# same-box A/B of two builds in track mode (1.25 M sequences): build/libplaac_old.so against the tree's library
mkdir -p gpurun_out/r4
timeout -k 10 120 python3 tools/write_rate_probe.py > gpurun_out/r4/write_rate_probe.json 2>gpurun_out/r4/write_rate_probe.err; cat gpurun_out/r4/write_rate_probe.json
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "track"  > gpurun_out/r4/tracks_pytest.txt 2>&1 || { tail -30 gpurun_out/r4/tracks_pytest.txt; exit 1; }
tail -2 gpurun_out/r4/tracks_pytest.txt
out=gpurun_out/r4/tracks_ab_builds.txt; : > $out
F="--steps 10 --warmup 3 --no-cpu-baseline --no-e2e --no-predict --no-clock-probe --no-host-leg --tracks --nprot 1250000"
for rep in 1 2 3; do for v in old new; do
  if [ $v = old ]; then export PLAAC_NATIVE_LIB=$PWD/build/libplaac_old.so; else unset PLAAC_NATIVE_LIB; fi
  timeout -k 10 300 python3 bench.py $F "$@" 2>>gpurun_out/r4/tracks_ab_builds.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', 'ms/step', d['ms_per_step'])" >> $out || echo "$v failed" >> $out
done; done
cat $out
