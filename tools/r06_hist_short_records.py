#!/usr/bin/env python3
"""k_hist on SHORT records (peptide-like FASTA: 8 - 30 residues; and 1 - 3 residues): the window of 64 record starts does not span
a 4 KiB group there. Milliseconds per pass and GB/s for the library named by PLAAC_NATIVE_LIB (default: the tree's)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from plaac_amd import native
from oracle import oracle_ctypes as oc
dev = torch.device("cuda:0")
ctx = native.Context(native.make_params())
rng = np.random.default_rng(5)
for name, lo, hi, nrec in (("peptides 8-30", 8, 31, 6_000_000), ("tiny 1-3", 1, 4, 40_000_000), ("proteins 50-1200", 50, 1200, 200_000),
                           ("mixed: 2 M tiny records, then proteins", 0, 0, 0)):
    if nrec:
        lens = rng.integers(lo, hi, nrec)
    else:
        lens = np.concatenate([rng.integers(1, 4, 2_000_000), rng.integers(50, 1200, 200_000)])
    offs = np.zeros(len(lens) + 1, dtype=np.uint64); offs[1:] = np.cumsum(lens)
    n = int(offs[-1])
    codes = rng.integers(1, 21, n, dtype=np.uint8)
    codes[rng.integers(0, n, n // 2000)] = 0
    want = oc.histogram(codes, offs)
    d_codes = torch.from_numpy(codes).to(dev); d_offs = torch.from_numpy(offs.astype(np.int64)).to(dev)
    cnt = torch.zeros(22, dtype=torch.int64, device=dev)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ctx.histogram_device(d_codes.data_ptr(), d_offs.data_ptr(), len(lens), cnt.data_ptr()); ctx.sync()
    ok = np.array_equal(cnt.cpu().numpy(), want)
    st = torch.cuda.Stream(dev)  # (a real stream: the library takes a null stream for 'its own')
    torch.cuda.synchronize()
    ev[0].record(st)
    for _ in range(3):
        ctx.histogram_device(d_codes.data_ptr(), d_offs.data_ptr(), len(lens), cnt.data_ptr(), stream=st.cuda_stream)
    ev[1].record(st); st.synchronize()
    ms = ev[0].elapsed_time(ev[1]) / 3
    print("%-40s %9d records %7.1f MB  %8.3f ms  %7.1f GB/s  %s" % (name, len(lens), n / 1e6, ms, (n + 8 * len(lens)) / ms / 1e6, "ok" if ok else "MISMATCH"), flush=True)
ctx.close()
