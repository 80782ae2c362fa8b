#!/bin/bash
# round 5: counters of the forward pass from the packed copy (k_fwd) and straight from the batch (k_fwd_direct), each by itself
# (the profiler serialises dispatches): where the direct form's time goes ("a committed A/B + PMC showing why not")
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r5/pmc_fd; rm -rf $O; mkdir -p $O
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-e2e --no-clock-probe --no-host-leg --no-predict --no-tracks-leg"
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
  "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY" \
  "GRBM_GUI_ACTIVE TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  for v in 0 1; do
    PLAAC_FWD_DIRECT=$v rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/s${i}_$v -- python3 bench.py $ARGS > /dev/null 2> $O/s${i}_$v.err || echo "FAILED set $i direct=$v"
  done
done
python3 - $O > gpurun_out/r5/pmc_fwd_direct.txt <<'PY'
import csv,glob,sys,collections,re
root=sys.argv[1]
tot=collections.defaultdict(lambda: collections.defaultdict(float)); calls=collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(root+'/*/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        m=re.search(r'::(k_fwd(?:_direct)?)[<(]', r['Kernel_Name'])
        if not m: continue
        tot[m.group(1)][r['Counter_Name']]+=float(r['Counter_Value']); calls[m.group(1)][r['Counter_Name']]+=1
print("# tools/r05_pmc_fwd_direct.sh: per launch over all 10 M sequences, each kernel by itself (FETCH_SIZE / WRITE_SIZE in KB; SQ_* in quad-cycles)")
names=sorted({c for k in tot for c in tot[k]})
print("%-30s %16s %16s %8s" % ("counter","k_fwd (packed)","k_fwd_direct","ratio"))
for x in names:
    a=tot['k_fwd'].get(x,0)/max(1,calls['k_fwd'].get(x,0)); b=tot['k_fwd_direct'].get(x,0)/max(1,calls['k_fwd_direct'].get(x,0))
    print("%-30s %16.5g %16.5g %8.2f" % (x,a,b,(b/a if a else float('nan'))))
PY
cat gpurun_out/r5/pmc_fwd_direct.txt; rm -rf $O/s*/
