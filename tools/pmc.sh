#!/bin/bash
# tools/pmc.sh - collect rocprofv3 PMC counters for the bench workload on the GPU box (one counter set per pass,
# never combined with sys/hip/hsa traces). Kernels are serialised (PLAAC_SERIAL_STREAMS=1) so counters attribute cleanly.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/pmc
export PLAAC_SERIAL_STREAMS=1
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/pmc/$tag -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmc/$tag.json 2> gpurun_out/pmc/$tag.err || echo "FAILED $tag"
done
ls -R gpurun_out/pmc | head -40
