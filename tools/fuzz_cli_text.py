#!/usr/bin/env python3
"""tools/fuzz_cli_text.py [iterations] [seed] - bin/plaac through the device's parser and formatter (default) against the same run
through the host's (PLAAC_DEVICE_PARSE=0 PLAAC_SINGLE_PASS=0: fastareader and the row formatter on the CPU, both held against
restatements of the reference's in the CPU tests): random FASTA files with every quirk fastareader has a rule for, random batch
cuts / contexts / options, stdout compared byte for byte. Prints one line per mismatch and a summary; exit status 1 on any."""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "bin", "plaac")
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
LETTERS = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", dtype=np.uint8)
ODD = np.frombuffer(b"acdxXBZ*- \t>.1\x80\xff", dtype=np.uint8)
TERMS = [b"\n", b"\r\n", b"\r"]


def fasta():
    out = []
    if rng.random() < 0.3:
        out.append(bytes(rng.choice(ODD, int(rng.integers(0, 30)))).replace(b">", b"x") + TERMS[int(rng.integers(0, 3))])
    nrec = int(rng.integers(0, 400)) if rng.random() < 0.9 else int(rng.integers(400, 3000))
    term_style = int(rng.integers(0, 4))  # one style for the file, or mixed
    for r in range(nrec):
        def t():
            return TERMS[int(rng.integers(0, 3))] if term_style == 3 else TERMS[term_style]
        name = b"rec%d" % r + bytes(rng.choice(ODD, int(rng.integers(0, 6)))).replace(b"\n", b"") + (b"  \t" if rng.random() < 0.2 else b"")
        out.append(b">" + name.replace(b"\r", b"") + t())
        n = int(rng.integers(0, 12)) if rng.random() < 0.15 else int(rng.integers(12, 900)) if rng.random() < 0.97 else int(rng.integers(900, 9000))
        seq = rng.choice(LETTERS, n)
        if rng.random() < 0.1 and n:
            k = rng.integers(0, n, max(1, n // 20))
            seq[k] = rng.choice(ODD, len(k))
        if rng.random() < 0.15 and n > 40:  # a PrD-like stretch
            a = int(rng.integers(0, n - 40))
            seq[a:a + int(rng.integers(40, min(n - a, 200) + 1))] = rng.choice(np.frombuffer(b"QNQNGYSQ", dtype=np.uint8), 1)[0]
        seq = bytes(seq).replace(b"\n", b"A").replace(b"\r", b"A")
        if rng.random() < 0.2:
            seq += b"*"
        width = int(rng.choice([60, 70, 80, 1 << 20, 13]))
        for i in range(0, len(seq), width):
            out.append(seq[i:i + width] + t())
        if rng.random() < 0.12:
            out.append(t() + b"skipped " + bytes(rng.choice(LETTERS, 20)) + t())
        if rng.random() < 0.03:
            out.append(b">" + t())
    data = b"".join(out)
    if rng.random() < 0.3 and data.endswith((b"\n", b"\r")):
        data = data.rstrip(b"\r\n")
    return data


bad = 0
with tempfile.TemporaryDirectory() as tmp:
    fa, o1, o2 = os.path.join(tmp, "f.fa"), os.path.join(tmp, "a.tsv"), os.path.join(tmp, "b.tsv")
    for it in range(iters):
        if it and it % 50 == 0:  # (a run on the GPU box must not stay silent for minutes)
            print("... %d iterations, %d mismatches so far" % (it, bad), flush=True)
        open(fa, "wb").write(fasta())
        args = ["-i", fa]
        if rng.random() < 0.3:
            args += ["-c", str(int(rng.choice([20, 30, 60, 90])))]
        if rng.random() < 0.3:
            args += ["-W", str(int(rng.choice([21, 41, 61])))]
        mode = rng.random()
        if mode < 0.25:
            args += ["-a", "0.5"]
        elif mode < 0.45:
            args += ["-a", str(rng.choice(["0.5", "1.0", "0.0"])), "-B", os.path.join(ROOT, "tests", "golden", "bg_freqs", "bg_freqs_HUMAN.txt")]
        if rng.random() < 0.15 and os.path.getsize(fa) < 400000:
            args += ["-p", "all"]  # (the per-residue table: device lines against the host's)
        if rng.random() < 0.2:
            args += ["-d"]
        if rng.random() < 0.2:
            args += ["-s"]
        env = dict(os.environ)
        env["PLAAC_BATCH_RECORDS"] = str(int(rng.choice([1, 7, 100, 262144])))
        env["PLAAC_BATCH_BYTES"] = str(int(rng.choice([100, 4096, 96 << 20])))
        if rng.random() < 0.3:
            env["PLAAC_DEVICES"] = "0,0"
        if rng.random() < 0.2:
            env["PLAAC_UPLOAD_THREAD"] = "0"
        host = dict(env, PLAAC_DEVICE_PARSE="0", PLAAC_SINGLE_PASS="0", PLAAC_DEVICE_FORMAT="0")
        to_file = rng.random() < 0.7
        outs = []
        for e, path in ((env, o1), (host, o2)):
            if to_file:
                with open(path, "wb") as fh:
                    r = subprocess.run([BIN] + args, stdout=fh, stderr=subprocess.PIPE, env=e, timeout=600)
                outs.append((r.returncode, open(path, "rb").read()))
            else:
                r = subprocess.run([BIN] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=e, timeout=600)
                outs.append((r.returncode, r.stdout))
        if outs[0] != outs[1]:
            bad += 1
            keep = os.path.join(ROOT, "gpurun_out", "fuzz_fail_%d_%d.fa" % (seed, it))
            os.makedirs(os.path.dirname(keep), exist_ok=True)
            open(keep, "wb").write(open(fa, "rb").read())
            print("MISMATCH iteration %d: args %s env %s rc %s/%s sizes %d/%d (input kept as %s)" % (
                it, args[2:], {k: v for k, v in env.items() if k.startswith("PLAAC_")}, outs[0][0], outs[1][0], len(outs[0][1]), len(outs[1][1]), keep))
print("fuzz_cli_text: %d iterations (seed %d), %d mismatches" % (iters, seed, bad))
sys.exit(1 if bad else 0)
