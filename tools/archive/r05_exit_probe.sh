#!/bin/bash
# round 5: where does bin/plaac's time go after its output is complete? (wall of the launcher vs the marks of PLAAC_TIMING_T0)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r5; mkdir -p $O
python3 - > $O/exit_probe.txt 2>&1 <<'PY'
import os, sys, subprocess, time
sys.path.insert(0, os.getcwd())
fa = "tests/golden/four_classic_prions.fasta"
def run(env, args, tag):
    for rep in range(3):
        t0 = time.clock_gettime_ns(time.CLOCK_MONOTONIC)
        r = subprocess.run(args, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=dict(os.environ, PLAAC_TIMING="1", PLAAC_TIMING_T0=str(t0), **env))
        t1 = time.clock_gettime_ns(time.CLOCK_MONOTONIC)
        marks = [l for l in r.stderr.decode().splitlines() if "after the launch" in l]
        print("%-40s wall %.1f ms | %s" % (tag, (t1 - t0) * 1e-6, " | ".join(m.split(":", 1)[1].split("(")[0].strip() for m in marks)))
run({}, ["bin/plaac", "-i", fa], "4 sequences")
run({"PLAAC_FAST_EXIT": "1"}, ["bin/plaac", "-i", fa], "4 sequences, _exit")
run({}, ["bin/plaac"], "usage only (no GPU)")
run({}, ["/bin/true"], "/bin/true")
PY
cat $O/exit_probe.txt
which strace perf ltrace 2>/dev/null
