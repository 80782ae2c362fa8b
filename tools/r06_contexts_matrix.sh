#!/bin/bash
# round 6: K independent jobs on one GPU under the host's two levers: the HIP runtime's hardware queues (GPU_MAX_HW_QUEUES,
# default 4 per priority class) and one stream per job (PLAAC_SERIAL_STREAMS=1)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r6; mkdir -p $O; out=$O/contexts_matrix.txt; : > $out
for cfg in ${CFGS:-2 3}; do
for spec in "default:" "hwq16:GPU_MAX_HW_QUEUES=16" "hwq32:GPU_MAX_HW_QUEUES=32" "serial:PLAAC_SERIAL_STREAMS=1" "serial+hwq32:PLAAC_SERIAL_STREAMS=1,GPU_MAX_HW_QUEUES=32" ${EXTRA}; do
  L=${spec%%:*}; E=${spec#*:}
  env X_=1 ${E//,/ } timeout -k 10 300 python3 tools/r06_contexts.py --config $cfg --contexts ${KS:-1 4 16} --steps 40 2>>$O/contexts_matrix.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('cfg$cfg %-14s' % '$L', '  '.join('K=%d: %.3f ms/step/job, %.2fe9 res/s (x%.2f)' % (l['contexts'], l['ms_per_step_per_job'], l['aggregate_residues_per_s']/1e9, l['vs_one_job']) for l in d['lines']))" >> $out || echo "cfg$cfg $L failed" >> $out
done; done
cat $out
