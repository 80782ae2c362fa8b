#!/usr/bin/env python3
"""debug: which bin/plaac configuration of test_streamed_pipeline hangs; stack of the hung process via rocgdb"""
import os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from plaac_amd import native, synth
P = native.make_params()
codes, offs = synth.make_batch(3, nprot=1500, seed=12, fg=np.array(P.fg), bg=np.array(P.bg), stop_fraction=0.1)
letters = np.frombuffer(b"XACDEFGHIKLMNPQRSTVWY*", dtype=np.uint8)[codes]
fa = "/tmp/hang.fa"
with open(fa, "wb") as fh:
    for i in range(len(offs) - 1):
        seq = letters[int(offs[i]):int(offs[i + 1])].tobytes()
        fh.write(b">rec%05d some description\n" % i)
        for k in range(0, len(seq), 60):
            fh.write(seq[k:k + 60] + b"\n")
envs = [{"PLAAC_BATCH_BYTES": 20000, "PLAAC_DEVICES": "0,0,0", "PLAAC_CLI_DEBUG": 1, "PLAAC_STREAM_DEBUG": 1},
        {"PLAAC_BATCH_BYTES": 20000, "PLAAC_DEVICES": "0,0,0", "PLAAC_OVERLAP_CALLS": 0},
        {"PLAAC_BATCH_BYTES": 20000, "PLAAC_DEVICES": "0,0,0", "PLAAC_MIXED": 0},
        {"PLAAC_BATCH_RECORDS": 100, "PLAAC_KEEP_BYTES": 1}, {"PLAAC_BATCH_RECORDS": 1, "PLAAC_DEVICES": "0,0"},
        {"PLAAC_BATCH_RECORDS": 100, "PLAAC_KEEP_BYTES": 100000, "PLAAC_CTX_PER_DEVICE": 3},
        {"PLAAC_FAST_EXIT": 1, "PLAAC_BATCH_RECORDS": 500}, {"PLAAC_TEARDOWN": 1, "PLAAC_BATCH_RECORDS": 300}]
for env in envs:
    e = dict(os.environ); e.update({k: str(v) for k, v in env.items()})
    t = time.time()
    p = subprocess.Popen([os.path.join(ROOT, "bin", "plaac"), "-i", fa, "-a", "0.5", "-c", "40"], stdout=open("/tmp/hang.out", "wb"),
                         stderr=subprocess.PIPE, env=e)
    try:
        p.wait(timeout=12)
        print(env, "rc", p.returncode, "%.2f s" % (time.time() - t), flush=True)
    except subprocess.TimeoutExpired:
        p.kill()
        err = p.stderr.read().decode(errors="replace")
        print(env, "HUNG after 25 s; stderr tail:\n" + "\n".join(err.splitlines()[-40:]), flush=True)
