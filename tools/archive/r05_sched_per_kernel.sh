#!/bin/bash
# round 5: per-kernel times (rocprofv3 kernel statistics, streams serialised) of the headline step under the default build and
# under a build with another scheduling strategy: which kernels like which
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r5; mkdir -p $O
V=${1:-max-ilp}
Q="--no-e2e --no-predict --no-tracks-leg --no-cpu-baseline --no-host-leg --no-clock-probe --steps 10 --warmup 2"
for lib in base $V; do
  L=$GRAFT_REPO_ROOT/plaac_amd/libplaac_native.so; [ $lib = base ] || L=$GRAFT_REPO_ROOT/plaac_amd/libplaac_native_$V.so
  rm -rf $O/prof_$lib
  PLAAC_NATIVE_LIB=$L PLAAC_SERIAL_STREAMS=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$lib -o p -- python3 bench.py $Q ${2:-} > $O/prof_$lib.json 2> $O/prof_$lib.err || echo "FAILED $lib"
done
python3 - <<PY
import csv, glob
def load(d):
    f = glob.glob("$O/prof_%s/**/*kernel_stats.csv" % d, recursive=True)[0]
    return {r["Name"]: (int(r["Calls"]), float(r["AverageNs"])) for r in csv.DictReader(open(f))}
a, b = load("base"), load("$V")
print("%-70s %8s %10s %10s %7s" % ("kernel", "calls", "base us", "$V us", "ratio"))
for k in sorted(a, key=lambda k: -a[k][0] * a[k][1])[:18]:
    if k in b:
        print("%-70s %8d %10.1f %10.1f %7.3f" % (k[:70], a[k][0], a[k][1] / 1e3, b[k][1] / 1e3, b[k][1] / a[k][1]))
PY
