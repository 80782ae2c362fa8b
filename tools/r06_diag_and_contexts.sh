#!/bin/bash
# round 6: (a) the tests of the diagnostic-only forms against the diagnostic build; (b) K independent jobs on one GPU, cfg2 / cfg3
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r6; mkdir -p $O
if [ "${SKIP_DIAG:-0}" != 1 ]; then
PLAAC_NATIVE_LIB=$GRAFT_REPO_ROOT/plaac_amd/libplaac_native_diag.so timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_schedule.py -x -q \
  -k "posteriors_from_the_forward_pass or segments or one_pass or window_kernel_in_input_order or chain_bound_sweep or schedule" > $O/diag_tests.txt 2>&1
tail -n 4 $O/diag_tests.txt
fi
for cfg in 2 3; do
  timeout -k 10 300 python3 tools/r06_contexts.py --config $cfg --contexts 1 2 4 8 16 --steps 40 > $O/contexts_cfg$cfg.json 2> $O/contexts_cfg$cfg.err || echo "cfg$cfg failed"
  cat $O/contexts_cfg$cfg.json
done
