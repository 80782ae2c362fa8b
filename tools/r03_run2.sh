cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r3; mkdir -p $O
timeout -k 10 600 build/issue_probe > $O/issue_probe.txt 2>&1; echo "probe rc=$?"
for i in 1 2 3; do timeout -k 10 300 python3 bench.py --no-e2e --no-cpu-baseline --steps 20 > $O/bench_rep$i.json 2>/dev/null; python3 -c "
import json;d=json.loads(open('$O/bench_rep$i.json').read().strip().splitlines()[-1]);print(d['ms_per_step'], d['roofline']['kernel_ms'])"; done
