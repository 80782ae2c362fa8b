#!/usr/bin/env python3
"""Large differential check of the summary-mode window kernel (filter tier + exact tier) against the oracle on the GPU box:
several million short proteins of four residue models (HMM-sampled, uniform, low-complexity, repeat-rich), every row
compared bit for bit; reports how many proteins each model sent to the exact tier.
    python tools/check_filter_large.py [proteins per model]  ->  gpurun_out/check_filter_large.txt"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle_ctypes as oc  # noqa: E402  (checker)
from plaac_amd import native, synth  # noqa: E402

n_per = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
P, Po = native.make_params(), oc.build_params()
rng = np.random.default_rng(99)
out = []


def batch(model):
    lens = rng.integers(30, 260, n_per)
    offs = np.zeros(n_per + 1, dtype=np.uint64)
    offs[1:] = np.cumsum(lens)
    total = int(offs[-1])
    if model == "hmm":
        codes, offs2 = synth.residues(lens, np.array(P.fg), np.array(P.bg), rng, stop_fraction=0.05)
        return codes, offs2
    if model == "uniform":
        return rng.integers(1, 21, total).astype(np.uint8), offs
    if model == "lowcomplexity":  # few residue types per protein: plateaus and near-ties in the smoothed tracks
        pal = rng.integers(1, 21, (n_per, 3))
        which = rng.integers(0, 3, total)
        rec = np.repeat(np.arange(n_per), lens)
        return pal[rec, which].astype(np.uint8), offs
    # repeats: every protein is a short unit repeated (exact ties between windows one period apart)
    period = rng.integers(2, 25, n_per)
    unit = rng.integers(1, 21, (n_per, 25))
    rec = np.repeat(np.arange(n_per), lens)
    pos = np.arange(total) - np.repeat(offs[:-1].astype(np.int64), lens)
    return unit[rec, pos % period[rec]].astype(np.uint8), offs


with native.Context(P) as ctx:
    for model in ("hmm", "uniform", "lowcomplexity", "repeats"):
        codes, offs = batch(model)
        t = time.time()
        got = ctx.score(codes, offs)
        tg = time.time() - t
        nfb = ctx.last_exact_fallbacks()
        with ctx.upload(codes, offs) as resident:  # device time of the scoring pass alone, residues resident
            for _ in range(3):
                resident.score()
            dev = ctx.last_timings(3)
        t = time.time()
        want = oc.score_batch(Po, codes, offs, nthreads=os.cpu_count() or 8)
        tc = time.time() - t
        same = got.tobytes() == want.tobytes()
        msg = ("%-14s %8d proteins %11d residues: rows identical to the oracle: %s; exact tier took %d (%.4f %%); device %.2f ms "
               "per pass (window kernels incl. exact tier %.2f ms); with upload %.2f s, oracle %.1f s") % (
            model, n_per, int(offs[-1]), same, nfb, 100.0 * nfb / n_per, dev["total"], dev["tracks"], tg, tc)
        print(msg, flush=True)
        out.append(msg)
        if not same:
            bad = np.nonzero(got.view(np.uint8).reshape(n_per, -1) != want.view(np.uint8).reshape(n_per, -1))[0]
            print("first differing proteins:", np.unique(bad)[:10])
            sys.exit(1)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
open(os.path.join(ROOT, "gpurun_out", "check_filter_large.txt"), "w").write("\n".join(out) + "\n")
