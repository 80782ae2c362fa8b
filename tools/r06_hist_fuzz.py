#!/usr/bin/env python3
"""Differential fuzz of k_hist (the background pass, countaas / isvalidprotein plaac.java:1698-1739) against the oracle on batches
of 20 - 80 MB - large enough for whole 4 KiB groups per wave, so that the window path, its fall-back to the ring inside a group,
skip_until over many groups and the single rows at the ends of a wave's range all occur: mixtures of record-length laws (runs of
empty / one-residue records, typical proteins, records of megabytes), X in bursts and at rates from 1e-6 to 0.5, stops and X as the
last residue, any alignment of the buffer start.   python3 tools/r06_hist_fuzz.py [iterations] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from plaac_amd import native
from oracle import oracle_ctypes as oc
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
dev = torch.device("cuda:0")
ctx = native.Context(native.make_params())
bad = 0
t0 = time.time()
for it in range(iters):
    target = int(rng.integers(20, 80)) << 20
    parts = []
    total = 0
    while total < target:  # segments of records drawn from one law each
        law = int(rng.integers(0, 6))
        k = int(rng.integers(1, 200000))
        if law == 0: lens = rng.integers(0, 3, k)                       # empty / one / two residues
        elif law == 1: lens = rng.integers(1, 40, k)                    # shorter than a window of 64 spans a group
        elif law == 2: lens = rng.integers(50, 1200, k // 4 + 1)        # proteins
        elif law == 3: lens = rng.integers(3000, 70000, k // 500 + 1)   # longer than a group
        elif law == 4: lens = rng.integers(1 << 20, 6 << 20, int(rng.integers(1, 3)))  # megabytes
        else: lens = np.full(k // 8 + 1, int(rng.integers(1, 5000)))    # all the same length
        parts.append(lens.astype(np.int64)); total += int(lens.sum())
    lens = np.concatenate(parts)
    offs = np.zeros(len(lens) + 1, dtype=np.uint64); offs[1:] = np.cumsum(lens)
    n = int(offs[-1])
    codes = rng.integers(1, 21, n, dtype=np.uint8)
    xr = 10.0 ** rng.uniform(-6, -0.3)
    codes[rng.integers(0, n, int(n * xr))] = 0
    for _ in range(int(rng.integers(0, 4))):  # bursts of X
        a = int(rng.integers(0, n)); codes[a:a + int(rng.integers(1, 20000))] = 0
    nz = np.nonzero(lens > 0)[0]
    last = (offs[1:][nz] - 1).astype(np.int64)
    codes[last[rng.random(len(last)) < rng.uniform(0, 0.6)]] = 21
    codes[last[rng.random(len(last)) < rng.uniform(0, 0.3)]] = 0
    first = offs[:-1][nz].astype(np.int64)
    codes[first[rng.random(len(first)) < rng.uniform(0, 0.2)]] = int(rng.choice([0, 21]))
    want = oc.histogram(codes, offs)
    shift = int(rng.integers(0, 64))
    buf = torch.zeros(n + 128, dtype=torch.uint8, device=dev)
    buf[shift:shift + n] = torch.from_numpy(codes).to(dev)
    d_offs = torch.from_numpy(offs.astype(np.int64)).to(dev)
    cnt = torch.zeros(22, dtype=torch.int64, device=dev)
    ctx.histogram_device(buf.data_ptr() + shift, d_offs.data_ptr(), len(lens), cnt.data_ptr())
    ctx.sync()
    ok = np.array_equal(cnt.cpu().numpy(), want)
    bad += not ok
    print("%3d: %8d records %6.1f MB  X rate %.1e  shift %2d  %s  (%.0f s)" % (it, len(lens), n / 1e6, xr, shift, "ok" if ok else "MISMATCH", time.time() - t0), flush=True)
    del buf, d_offs
print("r06_hist_fuzz: %d iterations (seed %d), %d mismatches" % (iters, seed, bad))
sys.exit(1 if bad else 0)
