#!/bin/bash
# tools/pmc_quick.sh - two SQ counter passes for the bench workload (serialised kernels)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; rm -rf gpurun_out/pmcq; mkdir -p gpurun_out/pmcq
export PLAAC_SERIAL_STREAMS=1
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_LDS_UNALIGNED_STALL"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/pmcq/$tag -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > gpurun_out/pmcq/$tag.json 2> gpurun_out/pmcq/$tag.err || echo "FAILED $tag"
done
python3 - <<'PY'
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/pmcq/*/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name']
        for key in ('k_tracks20','k_vit','k_fwd','k_win'):
            if key in k:
                agg[key][r['Counter_Name']].append(float(r['Counter_Value'])); break
for k in agg:
    print(k, ' '.join('%s=%.3g'%(c,sum(v)/len(v)) for c,v in sorted(agg[k].items())))
PY
