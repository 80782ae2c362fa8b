#!/bin/bash
# what the GPU box's host gives a process: cores, quota, and how the row formatter scales over threads
cd $GRAFT_REPO_ROOT; O=gpurun_out/r5; mkdir -p $O
{
echo "nproc: $(nproc)"; lscpu | grep -E "Model name|Thread|Core|Socket|^CPU\(s\)|MHz" 
echo "cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"; echo "cpuset: $(cat /sys/fs/cgroup/cpuset.cpus.effective 2>/dev/null)"
g++ -O2 -std=c++17 -Iinclude tools/format_bench.cpp -o /tmp/format_bench -Lplaac_amd -lplaac_native -Wl,-rpath,$PWD/plaac_amd -Wl,-rpath,/opt/rocm/lib || exit 1
echo "--- one process"; /tmp/format_bench 1000000 | tail -1
for n in 4 8 16 32; do
  echo "--- $n processes side by side"
  for i in $(seq $n); do /tmp/format_bench 1000000 | tail -1 & done | sort | sed -n '1p;$p'
  wait
done
} > $O/host_probe.txt 2>&1
cat $O/host_probe.txt
