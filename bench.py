#!/usr/bin/env python3
"""bench.py — PLAAC scoring hot path on MI355X: residues/s with roofline and CPU baseline.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--nprot P] [--config 2|3|4] [--tracks]
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Workload (BASELINE.json configs[3], the one `metric` is quoted on): UniRef50-shaped synthetic proteome,
10 M sequences sharded over 8 GPUs = 1.25 M sequences (~0.36 G residues) PER GPU, default parameters.
Scaling is weak: every rank scores its own 1.25 M-sequence shard, then the 160-byte summary rows are
gathered to rank 0 over RCCL (the only exchange of the path). A step = one pass of the whole hot path
(plan + recurrence kernel + window-track kernel + row gather) over the resident shard.
Inputs are resident in HBM before the timed region. Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0         # MI355X_MICROARCH.md: HBM3E spec 8 TB/s (6.3 TB/s achievable)
FP64_VALU_PEAK_GOPS = 39300.0  # 78.6 TFLOP/s vector fp64 counts FMA as 2; this path may not fuse -> 39.3 T op/s
ALGO_OPS_PER_RESIDUE = 470     # non-fusable fp64 ops per residue (summary mode)


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


def time_reference_jar(codes, offsets, nseq):
    """`java -jar $PLAAC_REF_JAR -i sample.fa` (second of two runs), or the reason it was skipped"""
    import shutil
    import subprocess
    import tempfile
    jar, java = os.environ.get("PLAAC_REF_JAR"), shutil.which("java")
    if not jar or not java or not os.path.exists(jar):
        return "skipped: no JVM on PATH" if not java else "skipped: PLAAC_REF_JAR not set"
    from plaac_amd import native
    off = offsets[:nseq + 1].cpu().numpy()
    text = native.decode(codes[:int(off[-1])].cpu().numpy())
    with tempfile.NamedTemporaryFile("w", suffix=".fa", delete=False) as fh:
        for i in range(nseq):
            fh.write(">s%d\n%s\n" % (i, text[int(off[i]):int(off[i + 1])]))
        path = fh.name
    try:
        dt = None
        for _ in range(2):
            t0 = time.perf_counter()
            subprocess.run([java, "-jar", jar, "-i", path], stdout=subprocess.DEVNULL, check=True, timeout=600)
            dt = time.perf_counter() - t0
        return {"value": round(int(off[-1]) / dt, 1), "unit": "residues/s", "cores": 1, "kind": "reference",
                "sample": "first %d sequences, parse+score+format, second of two runs" % nseq}
    finally:
        os.unlink(path)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", type=int, default=4, choices=(2, 3, 4))
    ap.add_argument("--nprot", type=int, default=0, help="sequences per GPU (default: config 4 -> 1,250,000)")
    ap.add_argument("--tracks", action="store_true", help="per-residue track mode (82 B/residue written)")
    ap.add_argument("--sweep", action="store_true", help="BASELINE config 5: a step = the 9-point sweep "
                    "alpha in {0,0.5,1} x core length in {30,60,90} over the resident shard (value counts every "
                    "residue once per point)")
    ap.add_argument("--naive-sweep", action="store_true", help="with --sweep: nine full passes instead of the "
                    "sweep-aware scheduler")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default=None, help="torch.distributed backend (default nccl = RCCL); "
                    "'gloo' + --one-device lets two ranks share one GPU for a plumbing check")
    ap.add_argument("--one-device", action="store_true", help="TEST ONLY: every rank uses cuda:0")
    ap.add_argument("--calibrate", action="store_true", help="after the timed region run the histogram kernel once "
                    "(it reads exactly R bytes): calibration of FETCH_SIZE for tools/pmc.sh")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from plaac_amd import dist as pdist
    from plaac_amd import native, synth

    rank, local_rank, world = pdist.init_process_group(args.backend)
    if args.one_device:
        local_rank = 0
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the scoring path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    nprot = args.nprot or {2: 5880, 3: 20600, 4: 1_250_000}[args.config]
    P = native.make_params()  # defaults: c=60, ww=41, alpha=1, fg28
    ctx = native.Context(P, device=local_rank)

    # ---- synthetic shard, generated in HBM (seed differs per rank) -------------------------------
    codes, offsets = synth.make_batch_torch(args.config, nprot, np.array(P.fg), np.array(P.bg), dev,
                                            seed=synth.SEED0 + args.config + 1000 * rank)
    total = int(offsets[-1].item())
    rows = torch.zeros(nprot * native.ROW_BYTES, dtype=torch.uint8, device=dev)
    d_tracks = None
    if args.tracks:
        trk = {k: torch.zeros(total, dtype=torch.uint8, device=dev) for k in native.TRACK_U8}
        trk.update({k: torch.zeros(total, dtype=torch.float64, device=dev) for k in native.TRACK_F64})
        d_tracks = {k: v.data_ptr() for k, v in trk.items()}
    gather_list = None
    if world > 1 and rank == 0:
        gather_list = [torch.empty_like(rows) for _ in range(world)]
    # N > 1: the row gather of step k (rank 0 <- every rank, RCCL) runs on its own stream and overlaps the kernels
    # of step k+1, which write the other of two row buffers; everything is drained inside the timed region
    rows_pp = [rows, torch.zeros_like(rows)] if world > 1 else [rows]
    comm = torch.cuda.Stream(dev) if world > 1 else None
    scored = [torch.cuda.Event() for _ in rows_pp]
    gathered = [torch.cuda.Event() for _ in rows_pp]
    step_no = [0]
    # a real (non-null) HIP stream: the kernels are launched on it, the HIP events that time them are
    # recorded on it, and RCCL orders the row gather after it
    stream = torch.cuda.Stream(dev)
    torch.cuda.synchronize(dev)

    sweep_params = None
    if args.sweep:
        cnt = torch.zeros(22, dtype=torch.int64, device=dev)
        ctx.histogram_device(codes.data_ptr(), offsets.data_ptr(), nprot, cnt.data_ptr())
        ctx.sync()
        counts = pdist.allreduce_counts(cnt.cpu().numpy(), device=dev if world > 1 else None)  # exchange (i)
        sweep_params = [native.make_params(alpha=a, corelength=c, bgcounts=counts.astype(np.float64))
                        for a in (0.0, 0.5, 1.0) for c in (30, 60, 90)]
    npoints = len(sweep_params) if sweep_params else 1

    sweep_rows = [torch.zeros_like(rows) for _ in range(npoints)] if sweep_params else None

    def step():
        with torch.cuda.stream(stream):
            if sweep_params and not args.naive_sweep:  # one planned pass, shared per-alpha work
                ctx.score_sweep_device(codes.data_ptr(), offsets.data_ptr(), nprot, total, sweep_params,
                                       [r.data_ptr() for r in sweep_rows], stream=stream.cuda_stream)
                if world > 1:
                    for r in sweep_rows:
                        dist.gather(r, gather_list, dst=0)
                return
            for k in range(npoints):
                if sweep_params:
                    ctx.set_params(sweep_params[k])
                b = step_no[0] % len(rows_pp)
                step_no[0] += 1
                if world > 1 and step_no[0] > len(rows_pp):
                    stream.wait_event(gathered[b])  # the gather that last read this buffer has finished
                ctx.score_device(codes.data_ptr(), offsets.data_ptr(), nprot, total, rows_pp[b].data_ptr(), d_tracks,
                                 stream=stream.cuda_stream)
                if world > 1:  # final gather of per-protein summary rows, ordered after this step's kernels
                    scored[b].record(stream)
                    with torch.cuda.stream(comm):
                        comm.wait_event(scored[b])
                        dist.gather(rows_pp[b], gather_list, dst=0)
                        gathered[b].record(comm)

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
        tot = torch.tensor([total, nprot], dtype=torch.int64, device=dev)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        job_res, job_prot = int(tot[0].item()), int(tot[1].item())
    else:
        job_res, job_prot = total, nprot

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

    if args.calibrate:
        cnt = torch.zeros(22, dtype=torch.int64, device=dev)
        ctx.histogram_device(codes.data_ptr(), offsets.data_ptr(), nprot, cnt.data_ptr(), stream=stream.cuda_stream)
        torch.cuda.synchronize(dev)
    if rank != 0:
        ctx.close()
        if world > 1:
            dist.destroy_process_group()
        return

    # ---- rank 0: kernel times (HIP events on each kernel's launch stream, mean over the timed steps) ----
    ktimes = ctx.last_timings(min(args.steps, 32))
    kern = {"k_vit": ktimes["vit"], "k_fwd": ktimes["fwd"], "k_win": ktimes["win"], "k_tracks": ktimes["tracks"]}
    if args.tracks:
        kern["k_bwd"] = ktimes["bwd"]
    dom = max(kern, key=kern.get)
    dom_ms = kern[dom]
    tb = native.TRACK_BYTES_PER_RESIDUE if args.tracks else 0
    # algorithmic bytes of ONE launch of each kernel (DESIGN.md "Algorithmic bytes"): codes R (1 B/residue)
    # + per protein 16 B of plan (8 offset + 4 length + 4 order) + the bytes of the 160 B row the kernel owns
    # [+ its per-residue track outputs in track mode]
    kbytes = {
        "k_vit": total * (1 + (1 if args.tracks else 0)) + nprot * (16 + 44),
        "k_fwd": total * (1 + (16 if args.tracks else 0)) + nprot * (16 + 8),  # track mode: a-pairs to scratch
        "k_bwd": total * (1 + 16) + nprot * 16,
        "k_win": total + nprot * (16 + 56),
        "k_tracks": total * (1 + (64 if args.tracks else 0)) + nprot * (16 + 52),
    }
    path_bytes = total * (1 + tb) + nprot * 168
    achieved = kbytes[dom] / (dom_ms * 1e-3) / 1e9
    # HBM traffic of the dominant kernel: PMC counters cannot be read from inside this process; tools/pmc.sh
    # collects FETCH_SIZE / WRITE_SIZE for this same workload and leaves the per-launch byte counts here
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as fh:
            tj = json.load(fh)
        if tj.get("workload") == [args.config, nprot, bool(args.tracks)]:
            traffic = tj["bytes_per_launch"].get(dom, tj["bytes_per_launch"].get(dom + "20"))
    except (OSError, ValueError, KeyError):
        pass
    path_ms = ktimes["total"]
    roofline = {
        "bound": "hbm", "kernel": dom, "achieved": round(achieved, 3), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBPS, 6), "traffic": traffic, "measured_copy_GBps": copy_gbps,
        "kernel_ms": {k: round(v, 4) for k, v in ktimes.items()},
        "kernels_overlap": "k_vit, k_fwd, k_win, k_tracks run concurrently on 4 HIP streams; total = first launch -> join",
        "path_achieved_GBps": round(path_bytes / (path_ms * 1e-3) / 1e9, 3),
        "note": "fp64-VALU-bound path (SURVEY 8d M3): secondary roof below",
        "valu_fp64": {
            "achieved_Gops": round(ALGO_OPS_PER_RESIDUE * total / (path_ms * 1e-3) / 1e9, 1),
            "peak_Gops": FP64_VALU_PEAK_GOPS,
            "frac": round(ALGO_OPS_PER_RESIDUE * total / (path_ms * 1e-3) / 1e9 / FP64_VALU_PEAK_GOPS, 4),
        },
    }

    # ---- rank 0: CPU baseline = the oracle (a port, not the Java reference: no JVM on this box) ----
    cpu = None
    if not args.no_cpu_baseline:
        from oracle import oracle_ctypes as oc
        Po = oc.build_params()
        nthreads = usable_cores()
        n_s = min(nprot, 250000)
        off_h = offsets[:n_s + 1].cpu().numpy().astype(np.uint64)
        codes_h = codes[:int(off_h[-1])].cpu().numpy()
        n_1 = min(n_s, 8000)
        oc.score_batch(Po, codes_h[:int(off_h[n_1])], off_h[:n_1 + 1], nthreads=1)  # scratch warm-up
        t1 = time.perf_counter()
        oc.score_batch(Po, codes_h[:int(off_h[n_1])], off_h[:n_1 + 1], nthreads=1)
        dt1 = time.perf_counter() - t1
        oc.score_batch(Po, codes_h, off_h, nthreads=nthreads)  # thread-pool and per-thread scratch warm-up
        t1 = time.perf_counter()
        want = oc.score_batch(Po, codes_h, off_h, nthreads=nthreads)
        dtn = time.perf_counter() - t1
        got = rows[:n_s * native.ROW_BYTES].cpu().numpy().view(native.ROW_DTYPE)
        cpu = {
            "value": round(int(off_h[-1]) / dtn, 1), "unit": "residues/s", "cores": nthreads, "kind": "port",
            "sample": "first %d sequences (%d residues) of rank 0's shard on %d OpenMP threads, second of two runs; "
                      "value_1core = first %d sequences on 1 thread; oracle/plaac_oracle.c restatement (not the Java "
                      "reference: no JVM on this box; its dead work omitted)" % (n_s, int(off_h[-1]), nthreads, n_1),
            "value_1core": round(int(off_h[n_1]) / dt1, 1),
            "gpu_rows_match_oracle": bool(got.tobytes() == want.tobytes()),
        }

    # when the operator supplies the real reference (a JVM on PATH and
    # PLAAC_REF_JAR=/path/plaac.jar) time it too, single-threaded as it is, on a small sample
    if cpu is not None:
        cpu["reference_jar"] = time_reference_jar(codes, offsets, min(nprot, 8000))

    out = {
        "metric": "residues/sec", "value": round(job_res * npoints * args.steps / dt, 1), "unit": "residues/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "proteins_per_sec": round(job_prot * npoints * args.steps / dt, 1),
        "config": {
            "workload": {2: "cfg2 yeast-shaped proteome", 3: "cfg3 human-shaped proteome",
                         4: "cfg4 UniRef50-shaped, 10M sequences over 8 GPUs = 1.25M sequences per GPU"}[
                args.config],
            "mode": ("tracks" if args.tracks else "summary") + (" x 9-point alpha/core sweep (cfg5)" if args.sweep
                                                                  else ""), "sequences_per_gpu": nprot,
            "residues_per_gpu": total, "sequences_total": job_prot, "residues_total": job_res,
            "params": "c=60 ww=41 alpha=1.0 fg=prd_freq_scer_28", "sharding": "by sequence, %d rank(s)" % world,
            "exchange": ("%s gather of 160 B rows to rank 0" % ("RCCL" if (args.backend or "nccl") == "nccl" else args.backend))
            if world > 1 else "none (1 GPU)",
        },
        "roofline": roofline,
        "cpu_baseline": cpu,
    }
    print(json.dumps(out), flush=True)
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
