#!/bin/bash
# tools/r04_evidence_pmc_full.sh - the PMC passes of the headline workload alone (part 1 of the evidence ran them with the
# predicted-scaling leg of bench.py inside: its 1.25 M-sequence calls were averaged into the per-launch means)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r4/ev; mkdir -p $O
echo "pmc cfg4 full"; bash tools/pmc.sh > $O/pmc_cfg4_full.log 2>&1
cp gpurun_out/pmc/summary.json $O/pmc_summary_cfg4_full.json; cp gpurun_out/pmc/pmc_traffic.json $O/pmc_traffic.json
cp $(find gpurun_out/pmc/trace_concurrent -name "*kernel_stats.csv" | head -1) $O/kernel_stats_concurrent_cfg4_full.csv
cp $(find gpurun_out/pmc/trace_serial -name "*kernel_stats.csv" | head -1) $O/kernel_stats_serial_cfg4_full.csv
python3 tools/timeline.py $(find gpurun_out/pmc/trace_concurrent -name "*kernel_trace.csv" | head -1) 3 2 > $O/timeline_concurrent_cfg4_full.txt
rm -rf gpurun_out/pmc
tail -3 $O/pmc_cfg4_full.log | cut -c1-300
