#!/usr/bin/env python3
"""tools/stress_overlap.py [rounds] [seed] - randomized stress of overlapping calls (plaac_ctx_set_overlap) on the GPU box:
rounds of 12 back-to-back calls on ONE context without a wait in between - random batch shapes (1 .. 30 k proteins, now and
then a protein of 10 k .. 70 k residues, empty records, an empty batch), random kinds (summary, track mode, 4-point sweep),
every call into buffers of its own - then one wait and every result against the oracle, bit for bit."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle_ctypes as oc  # noqa: E402  (checker)
from plaac_amd import native, synth  # noqa: E402
from test_gpu_parity import assert_rows_equal, assert_tracks_equal  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
P0, Po = native.make_params(), oc.build_params()
fg, bg = np.array(P0.fg), np.array(P0.bg)
dev = torch.device("cuda", 0)
sweep_pts = [native.make_params(alpha=a, corelength=c) for a in (1.0, 0.3) for c in (25, 70)]
RB = native.ROW_DTYPE.itemsize


def batch():
    kind = rng.integers(0, 10)
    if kind == 0:
        lens = np.array([], dtype=np.int64)
    elif kind < 3:
        lens = rng.integers(0, 400, int(rng.integers(1, 300)))
    elif kind < 8:
        lens = rng.integers(1, 700, int(rng.integers(300, 30000)))
    else:
        lens = np.concatenate([rng.integers(10000, 70001, int(rng.integers(1, 3))), rng.integers(1, 500, int(rng.integers(0, 3000)))])
    rng.shuffle(lens)
    if len(lens) == 0:
        return np.zeros(0, np.uint8), np.zeros(1, np.uint64)
    return synth.residues(lens, fg, bg, rng, stop_fraction=0.1)


t_start = time.time()
ncalls = 0
with native.Context(P0) as ctx:
    ctx.set_overlap(True)
    st = torch.cuda.Stream(dev)
    for r in range(rounds):
        calls = []
        for _ in range(12):
            c, o = batch()
            k = rng.choice(["sum", "sum", "sum", "sum", "trk", "swp"])
            if k == "trk" and int(o[-1]) > 3_000_000:
                k = "sum"
            calls.append((k, c, o))
        held = []
        for k, c, o in calls:  # uploads first: inputs are complete before the calls are made (the library's contract)
            dc = torch.from_numpy(np.ascontiguousarray(c)).to(dev) if len(c) else torch.zeros(16, dtype=torch.uint8, device=dev)
            do = torch.from_numpy(np.ascontiguousarray(o).view(np.int64)).to(dev)
            held.append((dc, do))
        # ... and so are the outputs: torch fills a new tensor on ITS current stream, and a buffer handed to an overlapping
        # call must not be in use by other pending work (the fill arriving after the scoring kernels' writes was the one
        # "mismatch" this tool ever found - in itself)
        pre = []
        for k, c, o in calls:
            n, tot = len(o) - 1, int(o[-1])
            if k == "swp":
                pre.append(([torch.zeros(max(n, 1), RB, dtype=torch.uint8, device=dev) for _ in sweep_pts], None))
                continue
            trk = None
            if k == "trk":
                trk = {t: torch.zeros(max(tot, 1), dtype=torch.uint8, device=dev) for t in native.TRACK_U8}
                trk.update({t: torch.full((max(tot, 1),), float("nan"), dtype=torch.float64, device=dev) for t in native.TRACK_F64})
            pre.append((torch.zeros(max(n, 1), RB, dtype=torch.uint8, device=dev), trk))
        torch.cuda.synchronize(dev)
        outs = []
        for (k, c, o), (dc, do), (rws, trk) in zip(calls, held, pre):
            n, tot = len(o) - 1, int(o[-1])
            if k == "swp":
                ctx.score_sweep_device(dc.data_ptr(), do.data_ptr(), n, tot, sweep_pts, [x.data_ptr() for x in rws], stream=st.cuda_stream)
                outs.append((rws, None))
                continue
            ctx.score_device(dc.data_ptr(), do.data_ptr(), n, tot, rws.data_ptr(),
                             None if trk is None else {t: v.data_ptr() for t, v in trk.items()}, stream=st.cuda_stream)
            outs.append((rws, trk))
        ctx.sync()
        torch.cuda.synchronize(dev)
        for i, ((k, c, o), (rws, trk)) in enumerate(zip(calls, outs)):
            n, tot = len(o) - 1, int(o[-1])
            what = "round %d call %d (%s, %d proteins, %d residues)" % (r, i, k, n, tot)
            if n == 0:
                continue
            if k == "swp":
                for Pn, x in zip(sweep_pts, rws):
                    want = oc.score_batch(oc.build_params(alpha=Pn.alpha, corelength=Pn.corelength), c, o, nthreads=16)
                    assert_rows_equal(x[:n].cpu().numpy().view(native.ROW_DTYPE).reshape(-1), want, what)
                continue
            want = oc.score_batch(Po, c, o, tracks=(k == "trk"), nthreads=16)
            if k == "trk":
                want, wtr = want
                assert_tracks_equal({t: v[:tot].cpu().numpy() for t, v in trk.items()}, wtr, c, o, what)
            assert_rows_equal(rws[:n].cpu().numpy().view(native.ROW_DTYPE).reshape(-1), want, what)
        ncalls += len(calls)
        print("round %d ok (%d calls so far, %.0f s)" % (r, ncalls, time.time() - t_start), flush=True)
print("stress_overlap: %d calls in %d rounds, all identical to the oracle (seed %d)" % (ncalls, rounds, seed))
