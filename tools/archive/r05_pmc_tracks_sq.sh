#!/bin/bash
# round 5: where do the waves of the track-mode kernels spend their cycles? SQ counters per kernel (rocprofv3 serialises the
# dispatches while it collects counters, so these are each kernel BY ITSELF), track mode, 1.25 M sequences, default forms
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r5/pmc_sq; rm -rf $O; mkdir -p $O
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-e2e --no-clock-probe --no-host-leg --no-predict --no-tracks-leg --tracks --nprot 1250000 $@"
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
  "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
  "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_INSTS_BRANCH SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64" \
  "GRBM_GUI_ACTIVE TCP_PENDING_STALL_CYCLES_sum TCC_EA0_WRREQ_STALL_sum TA_BUSY_avr TCP_TCC_WRITE_REQ_sum SQ_INSTS_VALU_MUL_F64"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/s$i -- python3 bench.py $ARGS > $O/s$i.json 2> $O/s$i.err || echo "FAILED set $i"
done
python3 - $O > gpurun_out/r5/pmc_tracks_sq.txt <<'PY'
import csv,glob,sys,collections,re
root=sys.argv[1]
tot=collections.defaultdict(lambda: collections.defaultdict(float)); calls=collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(root+'/*/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        m=re.search(r'::(k_\w+(?:<[^>]*>)?)', r['Kernel_Name'])
        if not m: continue
        tot[m.group(1)][r['Counter_Name']]+=float(r['Counter_Value']); calls[m.group(1)][r['Counter_Name']]+=1
print("# tools/r05_pmc_tracks_sq.sh: SQ counters per kernel and launch (each kernel by itself: the profiler serialises dispatches)")
names=sorted({c for k in tot for c in tot[k]})
for k in sorted(tot, key=lambda k:-tot[k].get('SQ_WAVE_CYCLES',0)):
    c=tot[k]; n=calls[k]
    per={x: c[x]/max(1,n[x]) for x in c}
    if per.get('SQ_WAVE_CYCLES',0) < 1e6: continue
    print(k)
    for x in names:
        if x in per: print("   %-28s %14.4g" % (x, per[x]))
PY
head -c 6000 gpurun_out/r5/pmc_tracks_sq.txt
rm -rf $O/s*/
