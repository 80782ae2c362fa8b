#!/bin/bash
# tools/r04_trace_all.sh <tag> [bench args] - every kernel of a short bench run (rocprofv3 --kernel-trace), compact
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4; mkdir -p $O
TAG=$1; shift
rm -rf $O/tn
rocprofv3 --kernel-trace --output-format csv -d $O/tn -- python3 bench.py --no-e2e --no-cpu-baseline --no-clock-probe --no-host-leg --no-predict --steps 8 --warmup 3 "$@" > $O/tn_$TAG.json 2> $O/tn.err
python3 - $(find $O/tn -name "*kernel_trace.csv" | head -1) > $O/trace_all_$TAG.txt <<'PY'
import csv,re,sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if 'anonymous namespace)::k_' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
t0=int(rows[0]['Start_Timestamp'])
for r in rows:
    name=re.sub(r"\(anonymous namespace\)::","",r["Kernel_Name"]); name=re.sub(r"^void ","",name).split("(")[0]
    print("q%-2s s%-3s %10.3f %10.3f %8.3f %s"%(r["Queue_Id"],r["Stream_Id"],(int(r["Start_Timestamp"])-t0)/1e6,(int(r["End_Timestamp"])-t0)/1e6,(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6,name[:50]))
PY
rm -rf $O/tn
tail -2 $O/tn_$TAG.json | cut -c1-300
