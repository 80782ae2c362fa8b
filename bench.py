#!/usr/bin/env python3
"""bench.py — PLAAC scoring hot path on MI355X: residues/s with roofline, CPU baseline and end-to-end leg.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config 2|3|4] [--nprot P] [--weak] [--tracks] [--sweep]
    N > 1: either under a launcher (python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...:
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment) or plainly `python bench.py --gpus N`: the parent then
    starts the N rank processes itself, as fresh children, before it has touched torch or HIP, relays their output and
    exits with the worst child status.

Workload (BASELINE.json configs[3], the one `metric` is quoted on): UniRef50-shaped synthetic proteome,
10 M sequences (~2.9 G residues), default parameters. It fits one GPU, so N = 1 scores ALL of it. At N > 1 the SAME
10 M-sequence proteome (every rank generates it from the same seeds) is cut by sequence over the ranks with
plaac_amd.dist.shard_plan (equal residue counts; 1.25 M sequences per GPU at N = 8 = BASELINE's "sharded 8 x MI355X"):
strong scaling, and the same line carries a `weak` object (every rank scores the whole proteome; `--weak` makes that the
headline instead). The 160-byte summary rows are gathered to rank 0 over RCCL and put back into input order there
(the only exchange of the path, inside the timed region) as 136-byte wire rows in blocks of their exact sizes
(include/plaac_native.h: rank 0 rebuilds the other 24 bytes from its own offsets and the core length).
A step = one pass of the whole hot path (plan + pack + recurrence kernels + window-track kernel [+ row
gather]) over the resident proteome (reference loop replaced: cli/src/plaac.java:755-948).
Inputs are resident in HBM before the timed region. Prints ONE JSON line on rank 0.
"""
import argparse
import hashlib
import json
import re
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0         # MI355X_MICROARCH.md: HBM3E spec 8 TB/s (6.3 TB/s achievable)
FP64_VALU_PEAK_GOPS = 39300.0  # 78.6 TFLOP/s vector fp64 counts FMA as 2; this path may not fuse -> 39.3 T op/s
ALGO_OPS_PER_RESIDUE = 470     # non-fusable fp64 ops per residue (summary mode)
SYNTH_CHUNK = 1_250_000        # the synthetic proteome is generated in HBM in pieces of this many sequences
CFG_NPROT = {2: 5880, 3: 20600, 4: 10_000_000}


def usable_cores():
    """host cores this process may actually use: affinity mask, capped by the cgroup CPU quota"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // p))
        except (OSError, ValueError):
            pass
    return n


def time_reference_jar(fasta_path, nres):
    """`java -jar $PLAAC_REF_JAR -i sample.fa` (second of two runs), or the reason it was skipped"""
    import shutil
    jar, java = os.environ.get("PLAAC_REF_JAR"), shutil.which("java")
    if not jar or not java or not os.path.exists(jar):
        return "skipped: no JVM on PATH" if not java else "skipped: PLAAC_REF_JAR not set"
    dt = None
    for _ in range(2):
        t0 = time.perf_counter()
        subprocess.run([java, "-jar", jar, "-i", fasta_path], stdout=subprocess.DEVNULL, check=True, timeout=900)
        dt = time.perf_counter() - t0
    return {"value": round(nres / dt, 1), "unit": "residues/s", "cores": 1, "kind": "reference",
            "sample": "%s, parse+score+format, second of two runs" % os.path.basename(fasta_path)}


HDR_ALPHABET = b"ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789 =_-"
HDR_MIN, HDR_MAX = 80, 160   # bytes of a 'uniref' header line incl. '>' (UniRef50 deflines: cluster id, name, n=, Tax=, TaxID=, RepID=)
WRAP = 60                    # residues per sequence line of the 'uniref' shape (the reference's own inputs: cli/example/Scer.fasta)


def header_len(i):
    """length law of the 'uniref' header lines: 80 .. 160 bytes incl. '>', a fixed function of the record index (numpy or torch)"""
    return HDR_MIN + (i * 37 + (i >> 5) * 11) % (HDR_MAX - HDR_MIN + 1)


def header_name(i):
    """SEQid of record i of the 'uniref' shape (the header line without '>'), rebuilt on the host for the table check"""
    n = int(header_len(np.int64(i))) - 1
    fixed = b"UniRef50_%010d " % i
    j = np.arange(len(fixed), n, dtype=np.int64)
    body = np.frombuffer(HDR_ALPHABET, np.uint8)[(i * 131 + j * 7 + (i >> 3)) % len(HDR_ALPHABET)].tobytes()
    return fixed + body[:-1] + b"x"  # (never ends in a blank: only the first record's name is trimmed by the reference, :4362)


def write_fasta(torch, codes, offsets, nseq, path, shape="single"):
    """FASTA text of the first nseq records of the resident batch, assembled in HBM and written with one tofile().
    shape 'single': 10-byte header '>s%07d\n', ONE sequence line per record (the least text a FASTA file can be);
    shape 'uniref': header lines of 80 - 160 bytes (header_len / header_name) and sequence lines wrapped at 60 columns, as
    the reference's own inputs and UniRef releases are (fastareader concatenates the lines, plaac.java:4325-4340; SEQid is
    the whole header, :805). Returns (bytes, residues)."""
    dev = codes.device
    off = offsets[:nseq + 1]
    nres = int(off[-1].item())
    lens = off[1:] - off[:-1]
    letters = torch.tensor(list(b"XACDEFGHIKLMNPQRSTVWY*"), dtype=torch.uint8, device=dev)
    idx = torch.arange(nseq, device=dev, dtype=torch.int64)
    if shape == "single":
        out = torch.full((nres + 11 * nseq,), 10, dtype=torch.uint8, device=dev)  # '\n' everywhere first
        rec = torch.repeat_interleave(idx, lens)
        dst = torch.arange(nres, device=dev, dtype=torch.int64) + 11 * rec + 10
        out[dst] = letters[codes[:nres].long()]
        del rec, dst
        hdr0 = off[:-1] + 11 * idx
        out[hdr0] = ord(">")
        out[hdr0 + 1] = ord("s")
        for d in range(7):
            out[hdr0 + 2 + d] = (48 + (idx // 10 ** (6 - d)) % 10).to(torch.uint8)
        out.cpu().numpy().tofile(path)
        return int(out.numel()), nres
    assert shape == "uniref"
    hl = header_len(idx)                                   # header bytes incl. '>' (the newline behind it is extra)
    rec_bytes = hl + 1 + lens + (lens + WRAP - 1) // WRAP  # header + '\n' + residues + one '\n' per sequence line
    rec0 = torch.cumsum(rec_bytes, 0) - rec_bytes
    nbytes = int((rec0[-1] + rec_bytes[-1]).item())
    out = torch.full((nbytes,), 10, dtype=torch.uint8, device=dev)
    # residues, in pieces (index arrays of 8 bytes per residue)
    step = 1 << 28
    for r0 in range(0, nres, step):
        r1 = min(nres, r0 + step)
        g = torch.arange(r0, r1, device=dev, dtype=torch.int64)
        rec = torch.searchsorted(off, g, right=True) - 1
        t = g - off[rec]
        out[rec0[rec] + hl[rec] + 1 + t + t // WRAP] = letters[codes[r0:r1].long()]
        del g, rec, t
    # headers: '>UniRef50_%010d ' then the filler of header_name()
    alph = torch.tensor(list(HDR_ALPHABET), dtype=torch.uint8, device=dev)
    FIX = 1 + len(b"UniRef50_%010d " % 0)
    out[rec0] = ord(">")
    for d, ch in enumerate(b"UniRef50_"):
        out[rec0 + 1 + d] = ch
    for d in range(10):
        out[rec0 + 10 + d] = (48 + (idx // 10 ** (9 - d)) % 10).to(torch.uint8)
    out[rec0 + 20] = ord(" ")
    for i0 in range(0, nseq, 1 << 21):
        i1 = min(nseq, i0 + (1 << 21))
        fl = hl[i0:i1] - FIX                               # filler bytes of these records
        ri = torch.repeat_interleave(idx[i0:i1], fl)
        fstart = torch.cumsum(fl, 0) - fl
        j = torch.arange(int(fl.sum().item()), device=dev, dtype=torch.int64) - torch.repeat_interleave(fstart, fl) + (FIX - 1)
        out[rec0[ri] + 1 + j] = alph[(ri * 131 + j * 7 + (ri >> 3)) % len(HDR_ALPHABET)]
        del ri, j, fstart, fl
    out[rec0 + hl - 1] = ord("x")
    out.cpu().numpy().tofile(path)
    return nbytes, nres


def run_e2e(torch, codes, offsets, nseq, shape="uniref", keep=False):
    """FASTA bytes in -> TSV bytes out through bin/plaac (the C++ host above the C ABI), wall clock around the
    whole process (HIP start-up, parse, upload, kernels, download, formatting, write). keep: the FASTA and the TSV of the
    last run stay on disk (paths under _fasta / _tsv) for the table check; the caller removes them."""
    exe = os.path.join(ROOT, "bin", "plaac")
    if not os.path.exists(exe):
        return {"skipped": "bin/plaac not built"}
    tmp = os.environ.get("TMPDIR", "/tmp")
    fa = os.path.join(tmp, "plaac_bench_%d_%s.fa" % (os.getpid(), shape))
    tsv = os.path.join(tmp, "plaac_bench_%d_%s.tsv" % (os.getpid(), shape))
    ok = False
    try:
        t0 = time.perf_counter()
        fbytes, nres = write_fasta(torch, codes, offsets, nseq, fa, shape)
        t_write = time.perf_counter() - t0
        best, runs, timing = None, [], []
        for _ in range(2):
            if os.path.exists(tsv):
                os.unlink(tsv)  # (truncating the previous run's 2 GB of output is the shell's time, not the program's)
            t0 = time.perf_counter()
            with open(tsv, "wb") as fh:
                r = subprocess.run([exe, "-i", fa], stdout=fh, stderr=subprocess.PIPE, timeout=1800)
            dt = time.perf_counter() - t0
            if r.returncode != 0:
                return {"error": "bin/plaac exit %d: %s" % (r.returncode, r.stderr.decode(errors="replace")[-400:])}
            runs.append(round(dt, 4))
            if best is None or dt < best:
                best, timing = dt, [l for l in r.stderr.decode(errors="replace").splitlines() if l.startswith("plaac-timing")]
        obytes = os.path.getsize(tsv)
        out = {"value": round(nres / best, 1), "unit": "residues/s", "proteins_per_sec": round(nseq / best, 1),
               "wall_s": runs, "sequences": nseq, "residues": nres, "fasta_bytes": fbytes, "tsv_bytes": obytes,
               "fasta_shape": shape, "fasta_write_s": round(t_write, 3),
               "what": "bin/plaac -i <FASTA> > <TSV>, whole process incl. HIP start-up, best of two runs, 1 GPU; FASTA text: "
                       + ("header lines of %d - %d bytes (UniRef-style deflines; SEQid = the whole header, plaac.java:805), "
                          "sequence lines wrapped at %d columns (fastareader joins them, :4325-4340)" % (HDR_MIN, HDR_MAX, WRAP)
                          if shape == "uniref" else "10-byte headers, ONE sequence line per record (the least text a FASTA "
                          "file can be: no line joins, 10 bytes of SEQid)")}
        if timing:  # PLAAC_TIMING=1 in the environment: the host's own stage clock of the best run
            out["stages"] = timing
        if keep:
            out["_fasta"], out["_tsv"] = fa, tsv
        ok = True
        return out
    finally:
        for p in (() if (keep and ok) else (fa, tsv)):
            try:
                os.unlink(p)
            except OSError:
                pass


def file_sha256(path):
    import hashlib
    h = hashlib.sha256()
    with open(path, "rb") as fh:
        for chunk in iter(lambda: fh.read(1 << 24), b""):
            h.update(chunk)
    return h.hexdigest()


def tsv_line_starts(path):
    """byte offset of every line of the file (+ the file size as the last entry)"""
    starts, base = [np.zeros(1, np.int64)], 0
    with open(path, "rb") as fh:
        for chunk in iter(lambda: fh.read(1 << 26), b""):
            nl = np.flatnonzero(np.frombuffer(chunk, np.uint8) == 10)
            starts.append(nl.astype(np.int64) + (base + 1))
            base += len(chunk)
    out = np.concatenate(starts)
    if out[-1] != base:  # (no newline at the very end)
        out = np.append(out, base)
    return out


def check_table(tsv, checks, name_of, corelength=60, ww2=41):
    """The table bin/plaac wrote against the oracle's rows: `checks` = [(first record, oracle rows, codes, offsets)], every
    record's line rebuilt from the ORACLE's row by the library's host formatter (plaac_format_summary_row_n - itself held
    against a restatement of java.util.Formatter in tests/test_host_io.py) under the name name_of(record) and compared, byte
    for byte, with the line of that record in the file. The synthetic records are all scored (none is empty), so record i
    is line i behind the column header. Returns (lines checked, [mismatching (first, count) slices], total table lines)."""
    from plaac_amd import hostio
    starts = tsv_line_starts(tsv)
    nlines = len(starts) - 1
    with open(tsv, "rb") as fh:
        head = fh.read(1 << 20)
    hl = head.split(b"\n")
    first = next(i for i, l in enumerate(hl) if l.startswith(b"SEQid\t"))
    checked, bad = 0, []
    with open(tsv, "rb") as fh:
        for (s, rows, codes, offs) in checks:
            n = len(rows)
            a, b = int(starts[first + 1 + s]), int(starts[first + 1 + s + n])
            fh.seek(a)
            got = fh.read(b - a)
            exp = hostio.format_summary_rows(rows, [name_of(s + i) for i in range(n)], codes, offs, corelength, ww2)
            checked += n
            if got != exp:
                bad.append([s, n])
    return checked, bad, nlines - first - 1


def check_slices(nfull):
    """(start, count) slices of the batch whose rows are re-scored by the oracle after the timed region: the first
    500 k sequences (also the CPU baseline's sample) and ten more slices of 50 k spread over the rest of the batch, the
    last one = the final 100 k sequences (residue offsets beyond 2^31 in the 10 M-sequence batch)."""
    first = min(nfull, 500_000)
    out = [(0, first)]
    if nfull >= first + 200_000:
        last = 100_000
        span = nfull - first - last
        out += [(first + (span - 50_000) * k // 8, 50_000) for k in range(9)]
        out.append((nfull - last, last))
    elif nfull > first:
        out.append((first, nfull - first))
    return out


DIAGNOSTIC_ENV = ("PLAAC_DEBUG_SKIP", "PLAAC_DEBUG_SKIP_FROM", "PLAAC_DEBUG_COUNTER", "PLAAC_VIT_STOP")


def launch_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: start the N ranks as fresh child processes (one per GPU, RANK /
    LOCAL_RANK / WORLD_SIZE / MASTER_* set), from a parent that has imported neither torch nor anything that touches HIP.
    Rank 0 prints the JSON line on the stdout it inherits. Returns the worst child status; when a rank fails the others
    are given a moment and then stopped by their exact PIDs (a rank waiting in a collective for a dead peer never ends)."""
    import signal
    import socket
    port = os.environ.get("MASTER_PORT")
    if not port:
        sk = socket.socket()
        sk.bind(("127.0.0.1", 0))
        port = str(sk.getsockname()[1])
        sk.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR=os.environ.get("MASTER_ADDR", "127.0.0.1"), MASTER_PORT=port, HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env))
    worst, failed_at = 0, None
    while any(p.poll() is None for p in procs):
        for p in procs:
            rc = p.poll()
            if rc is not None and rc != 0 and failed_at is None:
                failed_at = time.monotonic()
        if failed_at is not None and time.monotonic() - failed_at > 20:
            for p in procs:
                if p.poll() is None:
                    p.send_signal(signal.SIGTERM)
            failed_at = time.monotonic() + 1e9  # (once)
        time.sleep(0.2)
    for p in procs:
        rc = p.returncode
        worst = max(worst, rc if rc > 0 else (128 - rc if rc < 0 else 0))
    return worst


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=int, default=4, choices=(2, 3, 4))
    ap.add_argument("--nprot", type=int, default=0, help="sequences PER GPU, each rank its own (default: the whole "
                    "config - cfg4 = 10,000,000 - cut over the ranks; track mode 1,250,000 because 82 B/residue of "
                    "tracks for 10 M do not fit)")
    ap.add_argument("--weak", action="store_true", help="N > 1: headline = every rank scores its own whole proteome "
                    "(per-GPU work fixed) instead of ONE proteome cut over the ranks")
    ap.add_argument("--shard", action="store_true", help="(default since round 3; kept for old command lines)")
    ap.add_argument("--contexts", type=int, default=0, help="scoring contexts per GPU, used alternately so that "
                    "consecutive steps overlap (measured on MI355X: no gain - the contexts' role streams share the four "
                    "hardware queues of a priority class; default 1)")
    ap.add_argument("--tracks", action="store_true", help="per-residue track mode (82 B/residue written)")
    ap.add_argument("--sweep", action="store_true", help="BASELINE config 5: a step = the 9-point sweep "
                    "alpha in {0,0.5,1} x core length in {30,60,90} over the resident shard (value counts every "
                    "residue once per point)")
    ap.add_argument("--naive-sweep", action="store_true", help="with --sweep: nine full passes instead of the "
                    "sweep-aware scheduler")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true", help="skip the FASTA-in -> TSV-out leg through bin/plaac")
    ap.add_argument("--no-weak-leg", action="store_true", help="N > 1: skip the extra weak-scaling measurement")
    ap.add_argument("--e2e-nprot", type=int, default=0, help="sequences of the resident proteome written as FASTA "
                    "for the end-to-end leg (default: all of them, i.e. the 10 M sequences / 3.0 GB of cfg4)")
    ap.add_argument("--exchange", default="auto", choices=["auto", "rank0", "ranges"], help="N > 1, strong scaling: where the "
                    "rows of a step end up. rank0: wire rows gathered to rank 0 (a whole shard on each of its links); ranges: an "
                    "all-to-all after which every rank holds one contiguous range of the table (what a multi-process host "
                    "formats and writes at its own offset); auto: ranges from four ranks on")
    ap.add_argument("--backend", default=None, help="torch.distributed backend (default nccl = RCCL); "
                    "'gloo' + --one-device lets two ranks share one GPU for a plumbing check")
    ap.add_argument("--one-device", action="store_true", help="TEST ONLY: every rank uses cuda:0")
    ap.add_argument("--no-overlap", action="store_true", help="every step is ordered behind the whole previous step (default: "
                    "plaac_ctx_set_overlap - the planning and packing of a step run beside the last window kernels of the "
                    "step before it)")
    ap.add_argument("--no-predict", action="store_true", help="N = 1: skip the predicted strong-scaling leg (the 1/8 share "
                    "of the proteome timed by itself)")
    ap.add_argument("--no-clock-probe", action="store_true", help="skip the shader-clock measurement (extra untimed steps)")
    ap.add_argument("--no-host-leg", action="store_true", help="skip the PCIe-inclusive plaac_score calls on a 1.25 M-sequence "
                    "sample (counter passes: their launches would be averaged into the per-launch means)")
    ap.add_argument("--calibrate", action="store_true", help="after the timed region run the histogram kernel once "
                    "(it reads exactly R bytes): calibration of FETCH_SIZE for tools/pmc.sh")
    ap.add_argument("--max-len", type=int, default=0, help="clip every sequence length (cfg4 with 8192: the length law without its "
                    "0.01 %% tail of 8,193 - 36,000-residue records: a proteome whose step no single chain bounds)")
    ap.add_argument("--allow-diagnostics", action="store_true", help="run although PLAAC_DEBUG_* / PLAAC_VIT_STOP are set (only a "
                    "DIAG build of the library reads them, and its rows are wrong by design: tools/archive/r04_ablate*.sh)")
    ap.add_argument("--no-tolerance-leg", action="store_true", help="N = 1 default line: skip the `summary_value_tolerance` object")
    ap.add_argument("--no-tracks-leg", action="store_true", help="N = 1 default line: skip the `tracks` object (the 1.25 M-sequence "
                    "share scored in per-residue track mode - the HBM-bound regime of the path - in the same run)")
    args = ap.parse_args()
    set_diag = [k for k in DIAGNOSTIC_ENV if os.environ.get(k)]
    if set_diag and not args.allow_diagnostics:
        raise SystemExit("bench.py: %s set in the environment - result-breaking diagnostics; unset them or pass "
                         "--allow-diagnostics (the line then says so)" % ", ".join(set_diag))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))  # (before torch / HIP are touched in this process)
    # Sweeps run nine chains (three per alpha group) beside the window kernels: with the HIP runtime's default of four
    # hardware queues per priority class they share queues; INTEGRATION.md recommends twelve for sweep-heavy hosts (the
    # library then gives every group streams of its own). Must be set before the runtime initialises; the host's choice.
    if args.sweep:
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "12")

    import torch
    import torch.distributed as dist
    from plaac_amd import dist as pdist
    from plaac_amd import native, synth

    rank, local_rank, world = pdist.init_process_group(args.backend)
    t_start = time.perf_counter()
    if os.environ.get("PLAAC_BENCH_STACKS"):  # diagnostic: every rank's Python stack on stderr after that many seconds
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["PLAAC_BENCH_STACKS"]), exit=False)

    def mark(what):  # one short line per phase on stderr (rank 0): where a long multi-rank run is
        if rank == 0 and (world > 1 or os.environ.get("PLAAC_BENCH_PROGRESS")):
            print("bench.py [%6.1f s] %s" % (time.perf_counter() - t_start, what), file=sys.stderr, flush=True)
    if args.one_device:
        local_rank = 0
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the scoring path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    RB = native.ROW_BYTES

    strong = world > 1 and not args.weak and not args.nprot  # ONE proteome cut over the ranks
    nfull = args.nprot or (1_250_000 if (args.tracks and args.config == 4) else CFG_NPROT[args.config])
    two_pass = args.config == 3 and not args.sweep  # cfg3: -a 0.5, background from the scored input
    alpha = 0.5 if args.config == 3 else 1.0
    P = native.make_params()  # defaults: c=60, ww=41, alpha=1, fg28 (cfg3 replaces it inside every step)

    # ---- synthetic proteome, generated in HBM piece by piece. Strong scaling: the same seeds on every rank (every
    # rank builds the whole proteome and keeps its share); otherwise the seed differs per rank.
    seed_rank = 0 if strong else rank
    pieces, offs, base = [], [torch.zeros(1, dtype=torch.int64, device=dev)], 0
    mark("ranks joined; generating the proteome")
    for ci, start in enumerate(range(0, nfull, SYNTH_CHUNK)):
        c_, o_ = synth.make_batch_torch(args.config, min(SYNTH_CHUNK, nfull - start), np.array(P.fg), np.array(P.bg),
                                        dev, seed=synth.SEED0 + args.config + 1000 * seed_rank + 100000 * ci,
                                        max_len=args.max_len or None)
        pieces.append(c_)
        offs.append(o_[1:] + base)
        base += int(o_[-1].item())
    codes_full = torch.cat(pieces) if len(pieces) > 1 else pieces[0]
    offsets_full = torch.cat(offs)
    del pieces, offs
    total_full = int(offsets_full[-1].item())

    class Work:
        """one resident batch of this rank"""
        def __init__(self, codes, offsets, plans, nmax, sizes=None):
            self.codes, self.offsets = codes, offsets
            self.nprot, self.total = offsets.numel() - 1, int(offsets[-1].item())
            self.plans = plans  # strong scaling, rank 0: the input indices of every rank's rows (else None)
            self.nmax = nmax    # rows of the largest shard (row buffers; equal blocks for dist.gather in weak mode)
            self.sizes = sizes  # strong scaling: records of every rank's shard (exact block sizes of the wire-row gather)
            self.range_x = None  # strong scaling from four ranks on: the all-to-all that leaves every rank one range of the table

    use_ranges = world > 1 and not args.weak and not args.nprot and (args.exchange == "ranges" or (args.exchange == "auto" and world >= 4))
    mark("proteome in HBM: %d sequences" % nfull)
    if strong:
        all_plans = pdist.shard_plan_torch(offsets_full, world)  # the C partitioner (plaac_shard_plan), every rank the same
        mine = all_plans[rank]
        c_s, o_s = pdist.extract_shard_torch(codes_full, offsets_full, mine)
        plans = all_plans if rank == 0 else None
        sizes = [int(p_.numel()) for p_ in all_plans]
        range_x = pdist.RangeExchange(all_plans, offsets_full, rank, world, device=dev) if use_ranges else None
        del all_plans
        main_work = Work(c_s, o_s, plans, max(sizes), sizes)
        main_work.range_x = range_x
        del mine
    else:
        main_work = Work(codes_full, offsets_full, None, nfull)
    nctx = args.contexts or 1
    ctxs = [native.Context(P, device=local_rank) for _ in range(nctx)]
    # consecutive steps on the resident batch may overlap (plaac_ctx_set_overlap: the planning + packing of step k+1 beside
    # the scoring kernels of step k; the batch is complete before the first step, and at N > 1 - two row buffers alternate,
    # the gather reads one while the next step writes the other - the host waits for the gather that last read a buffer
    # before the buffer is handed to a step again); --no-overlap: every step behind the previous
    overlap = not args.no_overlap
    for c in ctxs:
        c.set_overlap(overlap)
    # real (non-null) HIP streams: the kernels are launched on them, the HIP events that time them are recorded on
    # them, and RCCL orders the row gather after them
    streams = [torch.cuda.Stream(dev) for _ in range(nctx)]
    comm = torch.cuda.Stream(dev) if world > 1 else None
    cnt = torch.zeros(22, dtype=torch.int64, device=dev)       # the job's counts (summed over the ranks): the two-pass parameters
    cnt_probe = torch.zeros(22, dtype=torch.int64, device=dev)  # the timed background pass over THIS rank's batch writes here

    def background_counts(W, ctx, stream):
        """pass 1 of the reference (plaac.java:377-384): histogram of the input, summed over the ranks"""
        ctx.histogram_device(W.codes.data_ptr(), W.offsets.data_ptr(), W.nprot, cnt.data_ptr(), stream=stream.cuda_stream)
        if world > 1:
            with torch.cuda.stream(stream):
                dist.all_reduce(cnt, op=dist.ReduceOp.SUM)  # exchange (i): 22 x int64
        stream.synchronize()
        return cnt.cpu().numpy()

    sweep_params = counts = None
    if args.sweep:
        counts = background_counts(main_work, ctxs[0], streams[0])
        sweep_params = [native.make_params(alpha=a, corelength=c, bgcounts=counts.astype(np.float64))
                        for a in (0.0, 0.5, 1.0) for c in (30, 60, 90)]
    npoints = len(sweep_params) if sweep_params else 1

    sweep_points = [(a, c) for a in (0.0, 0.5, 1.0) for c in (30, 60, 90)]

    def run_region(W, nsteps, nwarm, with_tracks, verify=False):
        """nwarm untimed + nsteps timed steps over W -> (seconds [max over ranks], buffers). verify: after the clock has
        stopped every output buffer is filled with 0xFF and ONE more (untimed) step runs - the buffers the caller then
        checks against the oracle are that step's, so a step that skipped work cannot hide behind an earlier one's rows."""
        nslots = max(nctx, 2 if world > 1 else 1)
        rows_pp = [torch.zeros(W.nmax * RB, dtype=torch.uint8, device=dev) for _ in range(nslots)]
        d_tracks = trk = None
        if with_tracks:
            trk = {k: torch.zeros(W.total, dtype=torch.uint8, device=dev) for k in native.TRACK_U8}
            trk.update({k: torch.zeros(W.total, dtype=torch.float64, device=dev) for k in native.TRACK_F64})
            d_tracks = {k: v.data_ptr() for k, v in trk.items()}
        # exchange (ii), strong scaling: 136-byte wire rows in blocks of their exact sizes, rank 0 rebuilds the rows from its
        # own offsets and puts them in input order; weak scaling (every rank its own proteome, whose offsets rank 0 does not
        # have): 160-byte rows in equal blocks
        wire_mode = world > 1 and W.sizes is not None
        gather_list = final = got = lens_of = None
        range_x = W.range_x
        if range_x is not None:  # every rank: the rows of its range of the table, per sweep point
            final = [torch.zeros(range_x.count, RB, dtype=torch.uint8, device=dev) for _ in range(npoints)]
        elif world > 1 and rank == 0:
            if wire_mode:
                got = [None] + [torch.empty(W.sizes[r] * pdist.WIRE_ROW_BYTES, dtype=torch.uint8, device=dev) for r in range(1, world)]
                lens_full = offsets_full[1:] - offsets_full[:-1]
                lens_of = [lens_full[W.plans[r]] for r in range(world)]
                final = [torch.zeros(nfull, RB, dtype=torch.uint8, device=dev) for _ in range(npoints)]
            else:
                gather_list = [torch.empty(W.nmax * RB, dtype=torch.uint8, device=dev) for _ in range(world)]
        lens_local = (W.offsets[1:] - W.offsets[:-1]) if wire_mode else None
        sweep_rows = [torch.zeros(W.nmax * RB, dtype=torch.uint8, device=dev) for _ in range(npoints)] if sweep_params else None
        scored = [torch.cuda.Event() for _ in range(nslots)]
        gathered = [torch.cuda.Event() for _ in range(nslots)]
        # the exchange timed by itself: events around every gather on the stream it runs on (the driver's 8-GPU curve then
        # decomposes into the scoring step and the RCCL gather)
        gather_t = [] if world > 1 else None
        step_no = [0]

        def gather(buf, point=0, corelength=60):
            """exchange (ii): rows of every rank -> rank 0 (RCCL over xGMI: one point-to-point link per peer), input order
            restored there"""
            if not wire_mode:
                dist.gather(buf, gather_list, dst=0)
                return
            if range_x is not None:  # (ii'): all-to-all, 1/N of a shard per link; every rank rebuilds the rows of its range
                range_x.exchange(buf, corelength, final[point])
                return
            host_staged = dist.get_backend() == "gloo"  # (plumbing runs: gloo moves device tensors at a crawl, 27 s per 680 MB)
            if rank != 0:
                wire = pdist.rows_to_wire_torch(buf[:W.nprot * RB], lens_local)
                for req in dist.batch_isend_irecv([dist.P2POp(dist.isend, wire.cpu() if host_staged else wire, 0)]):
                    req.wait()
                return
            into = [None] + [torch.empty(got[r].numel(), dtype=torch.uint8) if host_staged else got[r] for r in range(1, world)]
            for req in dist.batch_isend_irecv([dist.P2POp(dist.irecv, into[r], r) for r in range(1, world)]):
                req.wait()
            if host_staged:
                for r in range(1, world):
                    got[r].copy_(into[r])
            fin = final[point]
            fin[W.plans[0]] = buf.view(-1, RB)[:W.nprot]
            for r in range(1, world):
                fin[W.plans[r]] = pdist.rows_from_wire_torch(got[r], lens_of[r], corelength)

        def step():
            b = step_no[0] % nslots
            ctx, stream = ctxs[b % nctx], streams[b % nctx]
            step_no[0] += 1
            with torch.cuda.stream(stream):
                if world > 1 and step_no[0] > nslots:
                    if overlap:  # (two steps back: done long ago - the library's kernels do not follow stream-side waits here)
                        gathered[b].synchronize()
                    stream.wait_event(gathered[b])  # the gather that last read this buffer has finished
                if sweep_params and not args.naive_sweep:  # one planned pass, shared per-alpha work
                    ctx.score_sweep_device(W.codes.data_ptr(), W.offsets.data_ptr(), W.nprot, W.total, sweep_params,
                                           [r.data_ptr() for r in sweep_rows], stream=stream.cuda_stream)
                    if world > 1:
                        for k, r in enumerate(sweep_rows):
                            gather(r, k, sweep_points[k][1])
                    return
                if two_pass:  # cfg3: background pass, table setup (plaac.java:444-500) and upload are part of the step
                    ctx.set_params(native.make_params(alpha=alpha, bgcounts=background_counts(W, ctx, stream).astype(np.float64)))
                for k in range(npoints):
                    if sweep_params:
                        ctx.set_params(sweep_params[k])
                    ctx.score_device(W.codes.data_ptr(), W.offsets.data_ptr(), W.nprot, W.total, rows_pp[b].data_ptr(),
                                     d_tracks, stream=stream.cuda_stream)
                    if sweep_params:
                        sweep_rows[k].copy_(rows_pp[b], non_blocking=True)
                    if world > 1:  # final gather of per-protein summary rows, ordered after this step's kernels
                        scored[b].record(stream)
                        with torch.cuda.stream(comm):
                            comm.wait_event(scored[b])
                            g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                            g0.record(comm)
                            gather(rows_pp[b], k if sweep_params else 0, sweep_points[k][1] if sweep_params else P.corelength)
                            g1.record(comm)
                            gather_t.append((g0, g1))
                            gathered[b].record(comm)

        def fence():
            torch.cuda.synchronize(dev)
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize(dev)

        torch.cuda.synchronize(dev)  # (the buffers above were filled on torch's stream, the steps run on others)
        for _ in range(nwarm):
            step()
        fence()
        t0 = time.perf_counter()
        for _ in range(nsteps):
            step()
        fence()
        dt = time.perf_counter() - t0
        if world > 1:
            tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax.item())
        gms = None
        if gather_t:
            ts = [a.elapsed_time(b_) for a, b_ in gather_t[nwarm:]]
            gms = sum(ts) / max(1, len(ts))
        last = (step_no[0] - 1) % nslots
        if verify:  # the step that is checked: into poisoned buffers, after the clock has stopped
            for t_ in rows_pp + (sweep_rows or []) + (final or []) + (list(trk.values()) if trk else []):
                t_.fill_(0xFF) if t_.dtype == torch.uint8 else t_.view(torch.uint8).fill_(0xFF)
            torch.cuda.synchronize(dev)
            step()
            fence()
            last = (step_no[0] - 1) % nslots
        if range_x is not None:  # the checks read ONE table: the ranges brought to rank 0 after the clock has stopped
            torch.cuda.synchronize(dev)
            final = [pdist.gather_ranges(f_, nfull) for f_ in final]
            if rank != 0:
                final = None
        return dt, {"rows": rows_pp[last], "final": final, "gather_list": gather_list, "sweep_rows": sweep_rows, "trk": trk,
                    "gather_ms": gms, "verified_step": "one more untimed step into buffers filled with 0xFF" if verify else None}

    mark("shards cut; timed region")
    dt, bufs = run_region(main_work, args.steps, args.warmup, args.tracks, verify=True)
    mark("timed region done: %.3f s for %d steps; checked step made" % (dt, args.steps))
    nprot, total = main_work.nprot, main_work.total
    if world > 1:
        tot = torch.tensor([total, nprot], dtype=torch.int64, device=dev)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        job_res, job_prot = int(tot[0].item()), int(tot[1].item())
    else:
        job_res, job_prot = total, nprot
    # kernel times of the main region (HIP events on each kernel's launch stream, mean over the timed steps)
    per_ctx = max(1, min(args.steps * (npoints if args.naive_sweep else 1) // nctx, 32))
    kt = [c.last_timings(per_ctx) for c in ctxs]
    ktimes = {k: sum(t[k] for t in kt) / len(kt) for k in kt[0]}
    fallbacks = ctxs[0].last_exact_fallbacks() if not args.tracks else None

    # ---- N > 1, strong scaling: the weak-scaling figure beside it (every rank scores the WHOLE proteome it generated)
    weak = None
    if strong and not args.no_weak_leg and not args.tracks and not args.sweep:
        keep_final = bufs["final"]
        bufs["gather_list"] = bufs["rows"] = None
        mark("weak-scaling leg")
        wdt, wb = run_region(Work(codes_full, offsets_full, None, nfull), args.steps, 1, False)
        mark("weak-scaling leg done")
        del wb
        weak = {"value": round(total_full * world * args.steps / wdt, 1), "unit": "residues/s",
                "ms_per_step": round(wdt / args.steps * 1e3, 4), "scaling": "weak",
                "what": "every rank scores the whole %d-sequence proteome; rows gathered to rank 0" % nfull}
        bufs["final"] = keep_final
        torch.cuda.empty_cache()

    # achievable HBM copy rate on this box (SURVEY 8d M3 asks for it next to the nominal 8 TB/s): 1 GiB
    # device-to-device copy = 2 GiB of traffic, outside the timed region
    copy_gbps = None
    if rank == 0:
        a_ = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
        b_ = torch.empty_like(a_)
        b_.copy_(a_)
        torch.cuda.synchronize(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            b_.copy_(a_)
        e1.record()
        torch.cuda.synchronize(dev)
        copy_gbps = round(5 * 2 * (1 << 30) / (e0.elapsed_time(e1) * 1e-3) / 1e9, 1)
        del a_, b_

    hist_ms = None
    if rank == 0:  # the background pass alone (it is inside the step only for cfg3): HIP events on its launch stream
        st = streams[0]
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        with torch.cuda.stream(st):
            ctxs[0].histogram_device(main_work.codes.data_ptr(), main_work.offsets.data_ptr(), nprot, cnt_probe.data_ptr(), stream=st.cuda_stream)
            ev[0].record(st)
            for _ in range(3):
                ctxs[0].histogram_device(main_work.codes.data_ptr(), main_work.offsets.data_ptr(), nprot, cnt_probe.data_ptr(), stream=st.cuda_stream)
            ev[1].record(st)
        st.synchronize()
        hist_ms = ev[0].elapsed_time(ev[1]) / 3
        hist_sha = hashlib.sha256(cnt_probe.cpu().numpy().tobytes()).hexdigest()[:16]
    # the same step with every call ordered behind the whole previous one (plaac_ctx_set_overlap off): untimed extra steps
    by_itself = None
    if overlap and world == 1 and not args.tracks and not args.sweep:
        for c in ctxs:
            c.set_overlap(False)
        ndt, nb = run_region(main_work, max(3, min(args.steps, 10)), 1, False)
        del nb
        by_itself = {"ms_per_step": round(ndt / max(3, min(args.steps, 10)) * 1e3, 4),
                     "what": "plaac_ctx_set_overlap off: every step behind the whole previous step (extra untimed steps)"}
        for c in ctxs:
            c.set_overlap(True)
    # The same step with plaac_ctx_set_value_tolerance on (OFF in every other number of this line): the five floats at the PAPA
    # centre from sliding first-level sums - what the north star's own bar for floats (1e-6) buys. Its rows are checked below:
    # every index and every other field bit-identical to the oracle, the five floats within 1e-9.
    tol_leg = None
    if rank == 0 and world == 1 and not args.tracks and not args.sweep and not two_pass and not args.no_tolerance_leg:
        for c in ctxs:
            c.set_value_tolerance(True)
        tsteps_ = max(5, min(args.steps, 20))
        vdt, vb = run_region(main_work, tsteps_, 2, False, verify=True)
        for c in ctxs:
            c.set_value_tolerance(False)
        tol_leg = {"ms_per_step": round(vdt / tsteps_ * 1e3, 4), "residues_per_sec": round(total * tsteps_ / vdt, 1), "steps": tsteps_,
                   "what": "plaac_ctx_set_value_tolerance(1): PAPAcombo / PAPAprop / PAPAfi / PAPAllr / PAPAllr2 from first-level "
                           "window sums that slide over six neighbouring positions (k_refine_centres<., SLIDE>) instead of 41 "
                           "fixed-order taps per position; NOT the default and not in `value`", "_rows": vb["rows"]}
        del vb
    # What one GPU of an 8-GPU strong-scaling run would take: the share plaac_shard_plan deals rank 0 of 8 (1.25 M sequences
    # of the 10 M), timed like the main region (overlapping steps, no gather). efficiency = full step / (8 x share step):
    # the ceiling of the 8-GPU curve before any exchange - a single GPU can measure it, the curve itself needs the node.
    predicted = tracks_leg = None
    if rank == 0 and world == 1 and args.config == 4 and not args.nprot and not args.tracks and not args.sweep and not (
            args.no_predict and args.no_tracks_leg):
        idx8 = pdist.shard_plan_torch(offsets_full, 8, 0)
        c8, o8 = pdist.extract_shard_torch(codes_full, offsets_full, idx8)
        w8 = Work(c8, o8, None, idx8.numel())
        if not args.no_predict:
            sdt, sb = run_region(w8, max(args.steps, 20), 3, False)
            del sb
            share_ms = sdt / max(args.steps, 20) * 1e3
            predicted = {"n_gpus": 8, "share_sequences": int(idx8.numel()), "share_residues": w8.total,
                         "share_ms_per_step": round(share_ms, 4), "full_ms_per_step": round(dt / args.steps * 1e3, 4),
                         "efficiency": round(dt / args.steps * 1e3 / (8 * share_ms), 4),
                         "what": "rank 0's share of 8 (plaac_shard_plan) scored on this GPU, overlapping steps, no gather; "
                                 "efficiency = full-proteome step / (8 x share step): the ceiling of strong scaling before "
                                 "the row gather (extra untimed steps)"}
        # The HBM-bound regime of the path: per-residue track mode (plotsomefastas, plaac.java:587-649), 82 B
        # written per residue, on the same share, timed like the main region; the tracks of its last step - made after the
        # clock stopped, into buffers filled with 0xFF - are checked against the oracle below
        if not args.no_tracks_leg:
            tsteps = max(5, min(args.steps, 20))
            tdt, tbufs = run_region(w8, tsteps, 2, True, verify=True)
            t_ms = tdt / tsteps * 1e3
            t_bytes = w8.total * (1 + native.TRACK_BYTES_PER_RESIDUE) + w8.nprot * 168
            tracks_leg = {"workload": "rank 0's 1/8 share of the proteome (plaac_shard_plan: %d sequences, %d residues) in per-residue "
                                      "track mode, one GPU" % (w8.nprot, w8.total),
                          "steps": tsteps, "ms_per_step": round(t_ms, 4), "residues_per_sec": round(w8.total * tsteps / tdt, 1),
                          "algorithmic_bytes": t_bytes, "path_achieved_GBps": round(t_bytes / (t_ms * 1e-3) / 1e9, 1),
                          "peak_GBps": HBM_PEAK_GBPS, "frac": round(t_bytes / (t_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                          "what": "83 B/residue + 168 B/protein (SURVEY 8d M4) / wall-clock step (fence - steps - fence), "
                                  "inputs resident; frac = of the 8 TB/s HBM peak", "_bufs": tbufs, "_work": w8}
        else:
            del w8
        del c8, o8, idx8
        torch.cuda.empty_cache()
    # PCIe-inclusive rate (never `value`): the host-buffer entry point plaac_score on a bounded sample - residues and
    # offsets from host memory in, rows to host memory out, through the library's pinned staging
    host_io = None
    if rank == 0 and world == 1 and not args.tracks and not args.sweep and not args.no_host_leg:
        ns = min(nprot, 1250000)
        o_h = main_work.offsets[:ns + 1].cpu().numpy().astype(np.uint64)
        c_h = main_work.codes[:int(o_h[-1])].cpu().numpy()
        ctxs[0].sync()
        ctxs[0].score(c_h, o_h)
        t0h = time.perf_counter()
        ctxs[0].score(c_h, o_h)
        dth = time.perf_counter() - t0h
        host_io = {"value": round(int(o_h[-1]) / dth, 1), "unit": "residues/s", "sequences": ns, "residues": int(o_h[-1]),
                   "seconds": round(dth, 4), "bytes_in": int(o_h[-1]) + 8 * (ns + 1), "bytes_out": 160 * ns,
                   "what": "plaac_score (host buffers in, host rows out, pinned staging + PCIe + kernels), second of two "
                           "calls on the first %d sequences; PCIe-inclusive, never reported as `value`" % ns}
        del c_h, o_h
    # the shader clock the chip holds under this load (untimed extra steps; a sleeping wave beside the scoring kernels):
    # the issue roof below is priced in cycles
    clock = None
    if rank == 0 and world == 1 and not args.no_clock_probe:
        import threading
        try:
            idle = ctxs[0].clock_probe(5000)
            seen, stop = [], threading.Event()

            def prober():
                while not stop.is_set():
                    seen.append(ctxs[0].clock_probe(4000))

            th = threading.Thread(target=prober)
            th.start()
            run_region(main_work, max(5, min(args.steps, 20)), 1, args.tracks)
            stop.set()
            th.join()
            if seen:
                clock = {"idle_MHz": round(idle, 1), "under_load_MHz_mean": round(sum(seen) / len(seen), 1),
                         "under_load_MHz_min": round(min(seen), 1), "samples": len(seen),
                         "what": "plaac_clock_probe: s_sleep steps of a lone wave per 100 MHz tick, 4 ms windows during "
                                 "extra untimed steps of this workload"}
        except native.PlaacError as e:
            clock = {"error": str(e)}
    if args.calibrate:
        ctxs[0].histogram_device(main_work.codes.data_ptr(), main_work.offsets.data_ptr(), nprot, cnt_probe.data_ptr(), stream=streams[0].cuda_stream)
        ctxs[0].calibration_reads(main_work.codes.data_ptr(), total, stream=streams[0].cuda_stream)
        torch.cuda.synchronize(dev)
    if rank != 0:
        for c in ctxs:
            c.close()
        if world > 1:
            dist.destroy_process_group()
        return

    # ---- rank 0 ----
    kern = {"k_vit": ktimes["vit"], "k_fwd": ktimes["fwd"], "k_win": ktimes["win"], "k_tracks": ktimes["tracks"]}
    if args.tracks:
        kern["k_bwd"] = ktimes["bwd"]
    elif ktimes["bwd"] > 0.05:  # summary mode, mixed forms: slot 7 times the long run (k_long + Viterbi of the long groups + core search)
        kern["k_long"] = ktimes["bwd"]
    dom = max(kern, key=kern.get)
    dom_ms = kern[dom]
    tb = native.TRACK_BYTES_PER_RESIDUE if args.tracks else 0
    # ALGORITHMIC bytes of one pass: 1 B/residue code read + 8 B offset + 160 B row per protein
    # [+ 82 B/residue of tracks] [+ 1 B/residue for the background pass of cfg3]; `achieved` divides them by the
    # average launch duration of the longest kernel (the four scoring kernels overlap, see kernels_overlap)
    path_bytes = total * (1 + tb + (1 if two_pass else 0)) + nprot * 168
    # No one kernel carries the path's bytes (four to six kernel streams share every step), so the headline figure is the
    # STEP's: algorithmic bytes / wall-clock step (VERDICT r04 for track mode, r05 for summary mode). The longest kernel
    # stream stays beside it for the record.
    longest, longest_ms = dom, dom_ms
    dom, dom_ms = "step", dt / args.steps * 1e3
    achieved = path_bytes / (dom_ms * 1e-3) / 1e9
    # HBM traffic and executed instruction counts: PMC counters cannot be read from inside this process;
    # tools/pmc.sh collects them for this same workload and leaves the per-launch figures under profiles/
    traffic, traffic_all, exec_ops, issue_instr, traffic_cal, traffic_all_cal, fetch_cal, issue_classes = (None,) * 8
    stale_counters = None
    import glob
    for pth in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_traffic*.json"))):
        try:
            with open(pth) as fh:
                tj = json.load(fh)
            wkey = list(tj.get("workload") or [])
            if len(wkey) == 3:  # (files of rounds 1 - 4: no sweep flag)
                wkey.append(False)
            if wkey == [args.config, nprot, bool(args.tracks), bool(args.sweep)] and not args.max_len:  # (clipped lengths: another workload)
                sys.path.insert(0, os.path.join(ROOT, "tools"))
                import pmc_summary
                counters_sha, tree_sha = tj.get("kernels_source_sha16"), pmc_summary.kernels_sha()
                if counters_sha != tree_sha:  # counters of other kernel sources are not this code's: say so, quote nothing
                    stale_counters = {"file": os.path.basename(pth), "counters_for_sources": counters_sha, "tree_sources": tree_sha}
                    continue
                per = tj["bytes_per_launch"]
                # the kernels timed under `dom`: k_tracks = the window stream (filter + refine + exact tier, or the exact
                # stream kernel), k_vit = Viterbi + the core list
                names = [k for k in per if k.startswith(dom)]
                names += [k for k in per if (dom == "k_tracks" and k == "k_refine_centres") or
                          (dom == "k_vit" and k.startswith("k_core"))]
                traffic = sum(per[k] for k in names) if names else (tj.get("bytes_per_step") if dom == "step" else None)
                traffic_all = tj.get("bytes_per_step")
                perc = tj.get("bytes_per_launch_calibrated")
                if perc:  # read factor measured on this repo's own access shapes instead of the guide's x2 for all
                    traffic_cal = sum(perc[k] for k in names if k in perc)
                    traffic_all_cal = tj.get("bytes_per_step_calibrated")
                    fetch_cal = tj.get("fetch_calibration")
                exec_ops = tj.get("fp64_ops_per_residue_executed")
                issue_instr = tj.get("valu_lds_wave_instructions_per_step")
                issue_classes = tj.get("issue_model")
        except (OSError, ValueError, KeyError):
            pass
    if traffic_all is not None:  # (older rounds' files of the same workload are still under profiles/: not this line's concern)
        stale_counters = None
    # track mode: tools/pmc.sh serialises the streams, which selects the throughput forms of the chain kernels - the traffic of
    # the forms the library picks by itself comes from a separate pair of passes (tools/archive/r04_pmc_tracks_default_forms.sh)
    traffic_default_forms = None
    if args.tracks and nprot == 1250000 and not args.sweep and not args.max_len:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import pmc_summary
        for pth in sorted(glob.glob(os.path.join(ROOT, "profiles", "r0*_pmc_tracks_default_forms.txt"))):
            try:
                with open(pth) as fh:
                    txt = fh.read()
                sha = re.search(r"kernels_source_sha16: (\w+)", txt)
                tot = re.search(r"all scoring kernels: ([0-9.]+) GB per step", txt)
                if sha and tot and sha.group(1) == pmc_summary.kernels_sha():
                    traffic_default_forms = int(float(tot.group(1)) * 1e9)
            except (OSError, ValueError):
                pass
    path_ms = ktimes["total"]
    step_ms = dt / args.steps * 1e3
    # (rates are per timed step; with overlapping steps a call's own latency - first planning kernel to join - is longer)
    algo_gops = ALGO_OPS_PER_RESIDUE * total / (min(path_ms, step_ms) * 1e-3) / 1e9
    roofline = {
        "bound": "hbm", "kernel": dom, "achieved": round(achieved, 3), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBPS, 6), "traffic": traffic, "traffic_all_kernels": traffic_all,
        "what": "algorithmic bytes of one pass (SURVEY 8d M4) / wall-clock step (fence - steps - fence); traffic = counted "
                "HBM bytes of all kernels of a step",
        "longest_kernel_stream": {"name": longest, "ms": round(longest_ms, 4),
                                  "algorithmic_GBps": round(path_bytes / (longest_ms * 1e-3) / 1e9, 3)},
        "traffic_all_kernels_default_forms": traffic_default_forms,
        "traffic_calibrated": traffic_cal, "traffic_all_kernels_calibrated": traffic_all_cal,
        "fetch_calibration": fetch_cal,
        "algorithmic_bytes": path_bytes, "measured_copy_GBps": copy_gbps,
        "kernel_ms": {k: round(v, 4) for k, v in ktimes.items()},
        "kernels_overlap": "k_vit, k_fwd, k_win, k_tracks run concurrently on 4 HIP streams; total = first launch -> join"
                           + ("; chain-bound batch in mixed forms: bwd = the long run (latency forms of the long wave-groups on "
                              "two more streams), vit / fwd / win = the throughput-form runs" if (not args.tracks and ktimes["bwd"] > 0.05) else "")
                           + ("; consecutive steps overlap (plaac_ctx_set_overlap): the planning and packing of a step run "
                              "beside the last window kernels of the step before, so total (a call's latency) exceeds "
                              "ms_per_step" if (overlap and not args.tracks and not args.sweep) else "")
                           + ("; %d contexts alternate, consecutive steps overlap" % nctx if nctx > 1 else ""),
        "path_achieved_GBps": round(path_bytes / (step_ms * 1e-3) / 1e9, 3),
        "step_latency_ms": round(path_ms, 4),
        "histogram_pass": None if hist_ms is None else {
            "kernel": "k_hist", "ms": round(hist_ms, 4), "bytes": total + 8 * nprot, "counts_sha16": hist_sha,
            "achieved_GBps": round((total + 8 * nprot) / (hist_ms * 1e-3) / 1e9, 1),
            "frac_of_peak": round((total + 8 * nprot) / (hist_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
            "what": "background pass (countaas / isvalidprotein, plaac.java:1698-1739) over the resident batch, the one "
                    "kernel of the path that IS HBM-bound; mean of 3 launches by HIP events"},
        "note": "instruction-issue-bound path (SURVEY 8d M3): secondary roofs below",
        "valu_fp64": {
            "reference_equivalent": {"ops_per_residue": ALGO_OPS_PER_RESIDUE, "achieved_Gops": round(algo_gops, 1),
                                     "what": "the reference's own fp64 operation count x this rate; NOT executed - the "
                                             "filter tier decides from prefix sums"},
            "executed_pmc": None if not exec_ops else {
                "ops_per_residue": exec_ops, "achieved_Gops": round(algo_gops * exec_ops / ALGO_OPS_PER_RESIDUE, 1),
                "frac": round(algo_gops * exec_ops / ALGO_OPS_PER_RESIDUE / FP64_VALU_PEAK_GOPS, 4),
                "source": "SQ_INSTS_VALU_{ADD,MUL,FMA}_F64 x 64 lanes over all kernels of a step, profiles/pmc_traffic.json"},
            "peak_Gops": FP64_VALU_PEAK_GOPS,
        },
        # the roof that binds: wave-instructions x measured issue cost per class (tools/issue_probe.hip on MI355X,
        # profiles/r03_issue_probe.txt) on 1024 SIMDs at the 2.4 GHz peak clock; counts from the PMC passes of tools/pmc.sh
        "issue": None if not issue_instr else {
            "valu_lds_wave_instructions_per_step": issue_instr, "model": issue_classes,
            "ms_at_peak_issue": None if not issue_classes else round(issue_classes["cycles_per_step"] / 1024 / 2.4e9 * 1e3, 3),
            "frac": None if not issue_classes else round(issue_classes["cycles_per_step"] / 1024 / 2.4e9 * 1e3 / step_ms, 4),
            "shader_clock": clock,
            "frac_at_measured_clock": None if not (issue_classes and clock and clock.get("under_load_MHz_mean")) else round(
                issue_classes["cycles_per_step"] / 1024 / (clock["under_load_MHz_mean"] * 1e6) * 1e3 / step_ms, 4),
            "source": "SQ_INSTS_VALU (fp64 / other) + SQ_INSTS_LDS over all kernels of a step, profiles/*pmc_traffic*.json "
                      "collected for the kernel sources of this tree (sha checked); per-class costs + the class shares of this tree's listing: the newest profiles/r0*_issue_classes.json"},
        "pmc_counters_stale": stale_counters,
        "shader_clock": clock,
        "host_buffers_pcie_inclusive": host_io,
    }

    # ---- rank 0: CPU baseline = the oracle (a port, not the Java reference: no JVM on this box), and the parity check of
    # this run's rows: slices spread over the WHOLE batch (strong scaling: over the gathered, re-ordered table)
    mark("oracle check / cpu baseline")
    cpu, rc, oracle_keep = None, 0, []
    if not args.no_cpu_baseline:
        from oracle import oracle_ctypes as oc
        nthreads = usable_cores()
        if strong:
            chk_codes, chk_offs, chk_rows, chk_n = codes_full, offsets_full, bufs["final"][0].view(-1), nfull
        else:
            chk_codes, chk_offs, chk_rows, chk_n = main_work.codes, main_work.offsets, bufs["rows"], nprot
        if two_pass:  # same two-pass parameters as the GPU step (counts of the WHOLE input)
            Po = oc.build_params(alpha=alpha, bgcounts=cnt.cpu().numpy().astype(np.float64))
        else:
            Po = oc.build_params()

        def host_slice(s, n):
            o = chk_offs[s:s + n + 1].cpu().numpy().astype(np.int64)
            return chk_codes[int(o[0]):int(o[-1])].cpu().numpy(), (o - o[0]).astype(np.uint64), int(o[-1])

        slices = check_slices(chk_n)
        codes_h, off_h, _ = host_slice(*slices[0])
        n_s = slices[0][1]
        n_1 = min(n_s, 16000)
        oc.score_batch(Po, codes_h[:int(off_h[n_1])], off_h[:n_1 + 1], nthreads=1)  # scratch warm-up
        t1 = time.perf_counter()
        oc.score_batch(Po, codes_h[:int(off_h[n_1])], off_h[:n_1 + 1], nthreads=1)
        dt1 = time.perf_counter() - t1
        oc.score_batch(Po, codes_h, off_h, nthreads=nthreads)  # thread-pool and per-thread scratch warm-up
        t1 = time.perf_counter()
        want = oc.score_batch(Po, codes_h, off_h, nthreads=nthreads)
        dtn = time.perf_counter() - t1
        checked, max_off, bad = 0, 0, []
        oracle_keep = []  # (first record, oracle rows, codes, offsets) of every slice: the e2e leg checks the table with them
        if sweep_params:  # every point of the sweep against the oracle run with that point's parameters
            n_c = min(n_s, 20000)
            for k, (a, c) in enumerate((a, c) for a in (0.0, 0.5, 1.0) for c in (30, 60, 90)):
                Pk = oc.build_params(alpha=a, corelength=c, bgcounts=counts.astype(np.float64))
                wk = oc.score_batch(Pk, codes_h[:int(off_h[n_c])], off_h[:n_c + 1], nthreads=nthreads)
                gk = (bufs["final"][k].view(-1) if strong else bufs["sweep_rows"][k])[:n_c * RB].cpu().numpy()
                if gk.tobytes() != wk.tobytes():
                    bad.append(["sweep point", k])
            checked = 9 * n_c
        else:
            for (s, n) in slices:
                if s == 0:
                    w = want
                    max_off = max(max_off, int(off_h[-1]))
                    oracle_keep.append((s, w, codes_h, off_h))
                else:
                    c_h, o_h, end = host_slice(s, n)
                    w = oc.score_batch(Po, c_h, o_h, nthreads=nthreads)
                    max_off = max(max_off, end)
                    oracle_keep.append((s, w, c_h, o_h))
                got = chk_rows[s * RB:(s + n) * RB].cpu().numpy()
                if got.tobytes() != w.tobytes():
                    bad.append([s, n])
                checked += n
        match = not bad
        if tol_leg is not None:  # the tolerance leg's verified step: first slice against the oracle's rows, field by field
            trows = np.frombuffer(tol_leg.pop("_rows")[:n_s * RB].cpu().numpy().tobytes(), dtype=native.ROW_DTYPE)
            loose = ("papa_combo", "papa_prop", "papa_fi", "papa_llr", "papa_llr2")
            tbad, worst = [], 0.0
            for name in want.dtype.names:
                g_, w_ = trows[name], want[name]
                if name in loose:
                    fin = np.isfinite(w_)
                    same_special = np.array_equal(np.isnan(g_), np.isnan(w_)) and np.array_equal(np.isinf(g_), np.isinf(w_))
                    err = float((np.abs(g_[fin] - w_[fin]) / np.maximum(1.0, np.abs(w_[fin]))).max()) if fin.any() else 0.0
                    worst = max(worst, err)
                    if not same_special or err > 1e-9:
                        tbad.append(name)
                elif g_.dtype.kind == "f":
                    if not np.array_equal(g_.view(np.uint64), w_.view(np.uint64)):
                        tbad.append(name)
                elif not np.array_equal(g_, w_):
                    tbad.append(name)
            tol_leg.update(rows_checked=int(n_s), fields_outside_tolerance=tbad, max_relative_deviation_of_the_five_floats=worst,
                           tolerance="indices and every other field bit-identical to the oracle; the five floats within 1e-9 "
                                     "(relative, floor 1)", matches_oracle_within_tolerance=not tbad)
            if tbad:
                rc = 3
        if tracks_leg is not None:  # the track-mode leg: rows and all twelve tracks of its verified step against the oracle
            tw, tb_ = tracks_leg.pop("_work"), tracks_leg.pop("_bufs")
            n_t = min(tw.nprot, 4000)
            o_t = tw.offsets[:n_t + 1].cpu().numpy().astype(np.uint64)
            r_t = int(o_t[-1])
            c_t = tw.codes[:r_t].cpu().numpy()
            w_rows, w_trk = oc.score_batch(oc.build_params(), c_t, o_t, tracks=True, nthreads=nthreads)
            keep = np.ones(r_t, bool)
            lastpos = (o_t[1:][np.diff(o_t) > 0] - 1).astype(np.int64)
            keep[lastpos[c_t[lastpos] == 21]] = False  # (the entry of a trimmed stop is unspecified)
            tbad = []
            if tb_["rows"][:n_t * RB].cpu().numpy().tobytes() != w_rows.tobytes():
                tbad.append("rows")
            for k in native.TRACK_U8 + native.TRACK_F64:
                g = tb_["trk"][k][:r_t].cpu().numpy()[keep]
                w = w_trk[k][keep]
                if k in ("post0", "post1"):  # exp() on the device: rtol 1e-12 (every other track bit for bit)
                    ok = np.allclose(g, w, rtol=1e-12, atol=0)
                elif g.dtype.kind == "f":
                    ok = np.array_equal(np.isnan(g), np.isnan(w)) and np.array_equal(g.view(np.uint64)[~np.isnan(g)], w.view(np.uint64)[~np.isnan(w)])
                else:
                    ok = np.array_equal(g, w)
                if not ok:
                    tbad.append(k)
            tracks_leg.update(gpu_tracks_match_oracle=not tbad, mismatching_tracks=tbad, sequences_checked=n_t,
                              residues_checked_per_track=int(keep.sum()), verified_step=tb_["verified_step"],
                              tolerance="rows, vit, map and the eight window tracks bit-identical; post0 / post1 rtol 1e-12")
            if tbad:
                rc = 3
            del tw, tb_
        cpu = {
            "value": round(int(off_h[-1]) / dtn, 1), "unit": "residues/s", "cores": nthreads, "kind": "port",
            "sample": "first %d sequences (%d residues) of the proteome on %d OpenMP threads, second of two runs; "
                      "value_1core = first %d sequences on 1 thread; oracle/plaac_oracle.c restatement (not the Java "
                      "reference: no JVM on this box; its dead work omitted)" % (n_s, int(off_h[-1]), nthreads, n_1),
            "value_1core": round(int(off_h[n_1]) / dt1, 1),
            "gpu_rows_match_oracle": match,
            "rows_checked": checked, "slices_checked": [list(x) for x in slices] if not sweep_params else None,
            "max_residue_offset_checked": max_off, "mismatching_slices": bad,
            "checked_table": "gathered rows of all ranks in input order" if strong else "rows of rank 0",
        }
        if not match:
            rc = 3

    # ---- rank 0, N = 1: end to end through the C++ host (FASTA bytes in -> TSV bytes out), SURVEY 8(d) M1 ----
    # `e2e` = UniRef-shaped TEXT (long headers, 60-column lines) - the metric's "UniRef50-shaped FASTA"; `e2e_single_line` =
    # the same records as the least text a FASTA file can be (rounds 4 - 5 quoted this one). The uniref table is then checked
    # against the oracle over every slice the cpu_baseline leg scored (1.05 M records across all batches of the run, the last
    # 100 k among them) and, as a whole, by sha256 against one run of the host route (host parser + host formatter).
    e2e = e2e_single = None
    if world == 1 and not args.no_e2e and not args.tracks and not args.sweep:
        n_e2e = min(nprot, args.e2e_nprot or nprot)
        e2e_single = run_e2e(torch, main_work.codes, main_work.offsets, n_e2e, "single")
        e2e = run_e2e(torch, main_work.codes, main_work.offsets, n_e2e, "uniref", keep=True)
        fa, tsv = e2e.pop("_fasta", None), e2e.pop("_tsv", None)
        try:
            if tsv and cpu is not None and oracle_keep and not two_pass:
                checks = [(s_, w_, c_, o_) for (s_, w_, c_, o_) in oracle_keep if s_ + len(w_) <= n_e2e]
                t0 = time.perf_counter()
                nchk, tbad, nlines = check_table(tsv, checks, header_name, P.corelength, P.ww2)
                e2e.update(table_rows_checked=nchk, table_rows_match_oracle=not tbad, table_mismatching_slices=tbad,
                           tsv_lines=nlines, table_lines_expected=n_e2e, table_check_s=round(time.perf_counter() - t0, 2),
                           table_check="every record of the cpu_baseline slices (incl. the last 100 k): the oracle's row "
                                       "through the library's host formatter under the record's own header, byte for byte "
                                       "against that record's line of the TSV")
                if tbad or nlines != n_e2e:
                    rc = 3
            if tsv and fa:
                # the whole table once more through the host route (host FASTA parser, host formatter; same kernels)
                t0 = time.perf_counter()
                sha_dev = file_sha256(tsv)
                tsv2 = tsv + ".hostroute"
                env = dict(os.environ, PLAAC_DEVICE_PARSE="0", PLAAC_DEVICE_FORMAT="0")
                with open(tsv2, "wb") as fh:
                    t1 = time.perf_counter()
                    r = subprocess.run([os.path.join(ROOT, "bin", "plaac"), "-i", fa], stdout=fh, stderr=subprocess.PIPE, timeout=1800, env=env)
                    host_s = time.perf_counter() - t1
                if r.returncode == 0:
                    sha_host = file_sha256(tsv2)
                    e2e.update(table_sha256=sha_dev, table_sha_matches_host_route=sha_dev == sha_host,
                               host_route_wall_s=round(host_s, 4), sha_check_s=round(time.perf_counter() - t0, 2),
                               host_route="PLAAC_DEVICE_PARSE=0 PLAAC_DEVICE_FORMAT=0 bin/plaac -i <the same FASTA>: host parser "
                                          "and host formatter, one run")
                    if sha_dev != sha_host:
                        rc = 3
                else:
                    e2e["host_route_error"] = r.stderr.decode(errors="replace")[-300:]
                    rc = 3
                if os.path.exists(tsv2):
                    os.unlink(tsv2)
        finally:
            for p_ in (fa, tsv):
                if p_ and os.path.exists(p_):
                    os.unlink(p_)
        # when the operator supplies the real reference (a JVM on PATH and
        # PLAAC_REF_JAR=/path/plaac.jar) time it too, single-threaded as it is, on a bounded sample
        if cpu is not None and os.environ.get("PLAAC_REF_JAR"):
            fa = os.path.join(os.environ.get("TMPDIR", "/tmp"), "plaac_bench_ref_%d.fa" % os.getpid())
            try:
                _, nres_s = write_fasta(torch, main_work.codes, main_work.offsets, min(nprot, 8000), fa, "uniref")
                cpu["reference_jar"] = time_reference_jar(fa, nres_s)
            finally:
                if os.path.exists(fa):
                    os.unlink(fa)
    if cpu is not None and "reference_jar" not in cpu:
        cpu["reference_jar"] = time_reference_jar("", 0) if not os.environ.get("PLAAC_REF_JAR") else "skipped: no e2e leg"

    if tol_leg is not None:
        tol_leg.pop("_rows", None)
    if tracks_leg is not None:  # (--no-cpu-baseline: nothing was checked)
        tracks_leg.pop("_work", None)
        tracks_leg.pop("_bufs", None)
        tracks_leg.setdefault("gpu_tracks_match_oracle", None)
    wl = {2: "cfg2 yeast-shaped proteome (5,880 sequences)", 3: "cfg3 human-shaped proteome (20,600 sequences), "
          "-a 0.5 with the background counted from the input inside every step (two-pass)",
          4: "cfg4 UniRef50-shaped, 10M sequences"}[args.config]
    if args.max_len:
        wl += "; lengths clipped at %d" % args.max_len
    if args.nprot:
        wl += "; --nprot %d sequences per GPU" % nprot
    elif strong:
        wl += "; ONE proteome cut by sequence over %d GPUs (strong scaling, equal residue counts)" % world
    elif world > 1:
        wl += "; one whole proteome PER GPU (weak scaling, %d x the config)" % world
    else:
        wl += ("; 1.25M-sequence share in track mode (82 B/residue for 10M does not fit 288 GB)"
               if args.tracks and args.config == 4 else "; all of it on 1 GPU")
    out = {
        "metric": "residues/sec", "value": round(job_res * npoints * args.steps / dt, 1), "unit": "residues/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(step_ms, 4), "higher_is_better": True,
        "scaling": "weak" if (world > 1 and not strong) else "strong",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "proteins_per_sec": round(job_prot * npoints * args.steps / dt, 1),
        "config": {
            "workload": wl,
            "mode": ("tracks" if args.tracks else "summary") + (" x 9-point alpha/core sweep (cfg5)" if args.sweep
                                                                  else ""), "sequences_per_gpu": nprot,
            "residues_per_gpu": total, "sequences_total": job_prot, "residues_total": job_res,
            "params": "c=60 ww=41 alpha=%.1f fg=prd_freq_scer_28%s" % (alpha, " bg=input counts" if two_pass else ""),
            "sharding": "by sequence, %d rank(s)" % world, "contexts_per_gpu": nctx,
            "consecutive_steps_overlap": bool(overlap) and not args.tracks and not args.sweep,
            "hip_hardware_queues": os.environ.get("GPU_MAX_HW_QUEUES", "runtime default (4 per priority class)"),
            "step_by_itself": by_itself,
            "predicted_strong_scaling": predicted,
            "gather_ms_per_step": None if bufs.get("gather_ms") is None else round(bufs["gather_ms"], 4),
            "exchange": (("%s all-to-all of 136 B wire rows (include/plaac_native.h): every rank ends with one contiguous range of "
                          "the table, rows rebuilt and in input order (1/N of a shard per link; the ranges are brought to rank 0 "
                          "for the check after the clock has stopped)" % ("RCCL" if (args.backend or "nccl") == "nccl" else args.backend))
                         if use_ranges else
                         "%s gather to rank 0: %s" % ("RCCL" if (args.backend or "nccl") == "nccl" else args.backend,
                         "136 B wire rows (include/plaac_native.h), point-to-point blocks of their exact sizes, rows rebuilt and "
                         "put in input order on rank 0" if strong else "160 B rows in equal blocks (weak scaling)"))
            if world > 1 else "none (1 GPU)",
            "wire_row_bytes": pdist.WIRE_ROW_BYTES if strong else None,
            "verified_step": bufs.get("verified_step"),
            "diagnostics_in_environment": set_diag or None,
            "timed_region_s": round(dt, 3), "exact_tier_fallbacks_rank0": fallbacks,
        },
        "predicted_strong_efficiency": None if not predicted else predicted["efficiency"],
        "roofline": roofline,
        "tracks": tracks_leg,
        "summary_value_tolerance": tol_leg,
        "cpu_baseline": cpu,
        "e2e": e2e,
        "e2e_single_line": e2e_single,
    }
    if weak:
        out["weak"] = weak
    print(json.dumps(out), flush=True)
    for c in ctxs:
        c.close()
    if world > 1:
        dist.destroy_process_group()
    if rc:
        sys.exit(rc)


if __name__ == "__main__":
    main()
