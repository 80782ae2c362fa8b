#!/bin/bash
# HBM traffic of track mode in the forms the library picks by itself (tools/pmc.sh serialises the streams, which also selects
# the throughput forms): FETCH_SIZE / WRITE_SIZE passes without PLAAC_SERIAL_STREAMS, per-kernel sums per step
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r4/pmc_trk; rm -rf $O; mkdir -p $O
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-clock-probe --no-host-leg --no-predict --tracks --nprot 1250000"
for set in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/$set -- python3 bench.py $ARGS > $O/$set.json 2> $O/$set.err || echo "FAILED $set"
done
python3 - $O > gpurun_out/r4/pmc_tracks_default_forms.txt <<'PY'
import csv,glob,sys,collections,re
root=sys.argv[1]
tot=collections.defaultdict(lambda: collections.defaultdict(float)); steps=0
for f in glob.glob(root+'/*/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        m=re.search(r'::(k_\w+)', r['Kernel_Name'])
        if not m: continue
        tot[m.group(1)][r['Counter_Name']]+=float(r['Counter_Value'])
n=4.0  # 1 warm-up + 3 timed steps per pass
print("# tools/r04_pmc_tracks_default_forms.sh: track mode, 1.25 M sequences, default forms; GB per step = (2 x FETCH_SIZE + WRITE_SIZE) KB x 1024 / steps")
import os
sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import pmc_summary
print("# kernels_source_sha16: %s" % pmc_summary.kernels_sha())
s=0
for k,c in sorted(tot.items(), key=lambda kv:-(2*kv[1].get('FETCH_SIZE',0)+kv[1].get('WRITE_SIZE',0))):
    gb=(2*c.get('FETCH_SIZE',0)+c.get('WRITE_SIZE',0))*1024/1e9/n
    if k!='k_hist' and not k.startswith('k_calib') and not k.startswith('k_qprobe'): s+=gb
    print("%-20s read %7.2f  write %7.2f  total %7.2f GB"%(k, 2*c.get('FETCH_SIZE',0)*1024/1e9/n, c.get('WRITE_SIZE',0)*1024/1e9/n, gb))
print("all scoring kernels: %.1f GB per step"%s)
PY
cat gpurun_out/r4/pmc_tracks_default_forms.txt
rm -rf $O
