#!/bin/bash
# tools/r04_evidence_pmc.sh - round-4 evidence, part 1 (one gpurun call): rocprofv3 kernel traces + PMC passes of the headline
# workload (all 10 M sequences of cfg4) and of track mode (1.25 M-sequence share); summaries -> gpurun_out/r4/ev (copied to
# profiles/r04_* afterwards). Progress lines keep the call alive.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r4/ev; mkdir -p $O
echo "pmc cfg4 full"; bash tools/pmc.sh > $O/pmc_cfg4_full.log 2>&1
cp gpurun_out/pmc/summary.json $O/pmc_summary_cfg4_full.json; cp gpurun_out/pmc/pmc_traffic.json $O/pmc_traffic.json
cp $(find gpurun_out/pmc/trace_concurrent -name "*kernel_stats.csv" | head -1) $O/kernel_stats_concurrent_cfg4_full.csv
cp $(find gpurun_out/pmc/trace_serial -name "*kernel_stats.csv" | head -1) $O/kernel_stats_serial_cfg4_full.csv
python3 tools/timeline.py $(find gpurun_out/pmc/trace_concurrent -name "*kernel_trace.csv" | head -1) 3 2 > $O/timeline_concurrent_cfg4_full.txt
rm -rf gpurun_out/pmc
echo "pmc share (summary, mixed forms)"; bash tools/pmc.sh --nprot 1250000 > $O/pmc_share_1250k.log 2>&1
cp gpurun_out/pmc/summary.json $O/pmc_summary_share_1250k.json; cp gpurun_out/pmc/pmc_traffic.json $O/pmc_traffic_share_1250k.json
cp $(find gpurun_out/pmc/trace_concurrent -name "*kernel_stats.csv" | head -1) $O/kernel_stats_concurrent_share_1250k.csv
python3 tools/timeline.py $(find gpurun_out/pmc/trace_concurrent -name "*kernel_trace.csv" | head -1) 4 3 > $O/timeline_concurrent_share_1250k.txt
rm -rf gpurun_out/pmc
echo "pmc tracks"; bash tools/pmc.sh --tracks > $O/pmc_tracks_1250k.log 2>&1
cp gpurun_out/pmc/summary.json $O/pmc_summary_tracks_1250k.json; cp gpurun_out/pmc/pmc_traffic.json $O/pmc_traffic_tracks_1250k.json
cp $(find gpurun_out/pmc/trace_serial -name "*kernel_stats.csv" | head -1) $O/kernel_stats_serial_tracks_1250k.csv
python3 tools/timeline.py $(find gpurun_out/pmc/trace_concurrent -name "*kernel_trace.csv" | head -1) 2 > $O/timeline_concurrent_tracks_1250k.txt
rm -rf gpurun_out/pmc
ls -la $O
