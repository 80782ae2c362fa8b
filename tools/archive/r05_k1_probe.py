"""K1 probe: one 262144-record text batch through plaac_score_begin_text/_end_text vs the host parser + plaac_score_begin_counting"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bench
from plaac_amd import native, synth, hostio
dev = torch.device("cuda", 0)
P = native.make_params()
c_, o_ = synth.make_batch_torch(4, 262144, np.array(P.fg), np.array(P.bg), dev, seed=synth.SEED0 + 4)
fa = "/tmp/k1.fa"
fbytes, nres = bench.write_fasta(torch, c_, o_, 262144, fa)
print("# 262144 sequences, %d residues, %d bytes" % (nres, fbytes))
with native.Context(P) as ctx:
    for rep in range(4):
        t0 = time.perf_counter()
        for text, starts, trim in hostio.stream_fasta_text(fa, 1 << 20, 1 << 30):
            t1 = time.perf_counter()
            out = ctx.score_text(text, starts, counting=True)
            t2 = time.perf_counter()
        print("text: locate %.1f ms, begin+end %.1f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3))
        t0 = time.perf_counter()
        for text, starts, trim in hostio.stream_fasta_text(fa, 1 << 20, 1 << 30):
            t1 = time.perf_counter()
            table, lastb, cnt = ctx.score_text_table(text, starts, 60, 41, 1)
            t2 = time.perf_counter()
        print("text -> table: locate %.1f ms, begin + size + table %.1f ms, %d bytes of table" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, len(table)))
        t0 = time.perf_counter()
        names, codes, offs = hostio.read_fasta(fa)
        t1 = time.perf_counter()
        n = ctx.score_begin_counting(codes, offs)
        rows, cnt = ctx.score_end_counts(n)
        t2 = time.perf_counter()
        print("host: parse %.1f ms, begin+end %.1f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3))
