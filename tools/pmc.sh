#!/bin/bash
# tools/pmc.sh - rocprofv3 evidence for the bench workload, to be run on the GPU box:
#   1. kernel trace + stats of the default bench command (concurrent streams, as benchmarked)
#   2. kernel trace + stats with the four scoring kernels serialised (clean per-kernel durations)
#   3. PMC passes (one counter set per pass, never combined with sys/hip/hsa traces), kernels serialised
#   4. tools/pmc_summary.py -> <dir>/summary.json and pmc_traffic.json (copy both to profiles/)
#   PMC_DIR=<dir> (default gpurun_out/pmc): where everything goes. PMC_DEFAULT_FORMS=1: no PLAAC_SERIAL_STREAMS for the counter passes
#   (the profiler serialises the dispatches anyway while it collects counters): chain-bound workloads - config 3, the 1.25 M share,
#   sweeps over it, track mode - then run the kernel forms the library picks by itself instead of the throughput forms that the
#   serialised mode selects; the serial trace is skipped.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; D=${PMC_DIR:-gpurun_out/pmc}; rm -rf $D; mkdir -p $D
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-clock-probe --no-host-leg --no-predict --no-tracks-leg --no-tolerance-leg --calibrate $@"
rocprofv3 --kernel-trace --stats --output-format csv -d $D/trace_concurrent -- python3 bench.py $ARGS > $D/trace_concurrent.json 2> $D/trace_concurrent.err || echo "FAILED trace_concurrent"
if [ -z "$PMC_DEFAULT_FORMS" ]; then
export PLAAC_SERIAL_STREAMS=1
rocprofv3 --kernel-trace --stats --output-format csv -d $D/trace_serial -- python3 bench.py $ARGS > $D/trace_serial.json 2> $D/trace_serial.err || echo "FAILED trace_serial"
fi
for set in "FETCH_SIZE" "WRITE_SIZE" \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
  "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
  "GRBM_GUI_ACTIVE SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_LDS_UNALIGNED_STALL"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $D/$tag -- python3 bench.py $ARGS > $D/$tag.json 2> $D/$tag.err || echo "FAILED $tag"
done
python3 tools/pmc_summary.py $D
# (the traces are megabytes: keep the summaries and the per-kernel statistics)
for m in concurrent serial; do f=$(find $D/trace_$m -name '*kernel_stats.csv' 2>/dev/null | head -1); [ -n "$f" ] && cp $f $D/kernel_stats_$m.csv; done
rm -rf $D/trace_concurrent/*/ $D/trace_serial/*/ $D/FETCH_SIZE $D/WRITE_SIZE $D/SQ_WAVES $D/SQ_ACTIVE_INST_VALU $D/GRBM_GUI_ACTIVE 2>/dev/null
