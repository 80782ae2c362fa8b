#!/bin/bash
# tools/pmc.sh - rocprofv3 evidence for the bench workload, to be run on the GPU box:
#   1. kernel trace + stats of the default bench command (concurrent streams, as benchmarked)
#   2. kernel trace + stats with the four scoring kernels serialised (clean per-kernel durations)
#   3. PMC passes (one counter set per pass, never combined with sys/hip/hsa traces), kernels serialised
#   4. tools/pmc_summary.py -> gpurun_out/pmc/summary.json and pmc_traffic.json (copy both to profiles/)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; rm -rf gpurun_out/pmc; mkdir -p gpurun_out/pmc
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-clock-probe --no-host-leg --no-predict --calibrate $@"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pmc/trace_concurrent -- python3 bench.py $ARGS > gpurun_out/pmc/trace_concurrent.json 2> gpurun_out/pmc/trace_concurrent.err || echo "FAILED trace_concurrent"
export PLAAC_SERIAL_STREAMS=1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pmc/trace_serial -- python3 bench.py $ARGS > gpurun_out/pmc/trace_serial.json 2> gpurun_out/pmc/trace_serial.err || echo "FAILED trace_serial"
for set in "FETCH_SIZE" "WRITE_SIZE" \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
  "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
  "GRBM_GUI_ACTIVE SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_LDS_UNALIGNED_STALL"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/pmc/$tag -- python3 bench.py $ARGS > gpurun_out/pmc/$tag.json 2> gpurun_out/pmc/$tag.err || echo "FAILED $tag"
done
python3 tools/pmc_summary.py gpurun_out/pmc
