#!/usr/bin/env python3
"""round 6: what do K independent jobs in flight deliver on ONE GPU for the chain-bound BASELINE configs?
The web application's load is one `java -jar plaac.jar` per job (web/lib/server.rb:152-155, :533-541): K jobs = K host threads
here, each with its own scoring context, its own HIP stream and its own resident proteome (cfg2: 5,880 sequences, defaults;
cfg3: 20,600 sequences, -a 0.5 with the background counted from the input inside every step - two-pass), each running
`steps` steps back to back. Aggregate residues/s over the wall clock from a common start to the last thread's end.
    python3 tools/r06_contexts.py --config 3 --contexts 1 2 4 8 16 [--steps 40] [--overlap]"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=3, choices=(2, 3))
    ap.add_argument("--contexts", type=int, nargs="+", default=[1, 2, 4, 8, 16])
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--overlap", action="store_true", help="plaac_ctx_set_overlap on every context")
    args = ap.parse_args()
    import torch
    from plaac_amd import native, synth
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    P = native.make_params()
    nprot = {2: 5880, 3: 20600}[args.config]
    two_pass = args.config == 3
    out = {"config": args.config, "sequences": nprot, "two_pass": two_pass, "steps_per_job": args.steps, "overlap": args.overlap,
           "hip_hardware_queues": os.environ.get("GPU_MAX_HW_QUEUES", "runtime default"), "lines": []}
    kmax = max(args.contexts)
    jobs = []
    for k in range(kmax):  # every job its own proteome (seeded differently), context, stream, buffers
        c_, o_ = synth.make_batch_torch(args.config, nprot, np.array(P.fg), np.array(P.bg), dev, seed=synth.SEED0 + args.config + 17 * k)
        ctx = native.Context(P, device=0)
        ctx.set_overlap(args.overlap)
        jobs.append({"codes": c_, "offs": o_, "total": int(o_[-1].item()), "ctx": ctx, "stream": torch.cuda.Stream(dev),
                     "rows": torch.zeros(nprot * native.ROW_BYTES, dtype=torch.uint8, device=dev),
                     "cnt": torch.zeros(22, dtype=torch.int64, device=dev)})
    torch.cuda.synchronize(dev)

    def run_job(j, steps, gate):
        ctx, st = j["ctx"], j["stream"]
        gate.wait()
        for _ in range(steps):
            if two_pass:
                ctx.histogram_device(j["codes"].data_ptr(), j["offs"].data_ptr(), nprot, j["cnt"].data_ptr(), stream=st.cuda_stream)
                st.synchronize()
                ctx.set_params(native.make_params(alpha=0.5, bgcounts=j["cnt"].cpu().numpy().astype(np.float64)))
            ctx.score_device(j["codes"].data_ptr(), j["offs"].data_ptr(), nprot, j["total"], j["rows"].data_ptr(), None, stream=st.cuda_stream)
        st.synchronize()

    for K in args.contexts:
        for rep in range(2):  # (the first repetition warms up)
            gate = threading.Event()
            ths = [threading.Thread(target=run_job, args=(jobs[k], args.steps if rep else 3, gate)) for k in range(K)]
            for t in ths:
                t.start()
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            gate.set()
            for t in ths:
                t.join()
            torch.cuda.synchronize(dev)
            dt = time.perf_counter() - t0
        res = sum(jobs[k]["total"] for k in range(K)) * args.steps
        out["lines"].append({"contexts": K, "wall_s": round(dt, 4), "ms_per_step_per_job": round(dt / args.steps * 1e3, 4),
                             "aggregate_residues_per_s": round(res / dt, 1), "jobs_per_s": round(K * args.steps / dt, 2)})
        print(json.dumps(out["lines"][-1]), file=sys.stderr, flush=True)
    # K same-parameter jobs as ONE call: the host concatenates the jobs' records (rows come back in input order, so the split by
    # job is an offset) - what a job server can do for every job that runs with the default tables (alpha = 1: the tables do not
    # depend on the input)
    if not two_pass:
        for K in args.contexts:
            codes = torch.cat([jobs[k]["codes"] for k in range(K)])
            offs, basep = [jobs[0]["offs"][:1]], 0
            for k in range(K):
                offs.append(jobs[k]["offs"][1:] + basep)
                basep += jobs[k]["total"]
            offs = torch.cat(offs)
            rows = torch.zeros(K * nprot * native.ROW_BYTES, dtype=torch.uint8, device=dev)
            ctx, st = jobs[0]["ctx"], jobs[0]["stream"]
            torch.cuda.synchronize(dev)
            for rep in range(2):
                t0 = time.perf_counter()
                for _ in range(args.steps if rep else 3):
                    ctx.score_device(codes.data_ptr(), offs.data_ptr(), K * nprot, basep, rows.data_ptr(), None, stream=st.cuda_stream)
                st.synchronize()
                dt = time.perf_counter() - t0
            out["lines"].append({"contexts": 1, "jobs_merged_into_one_call": K, "wall_s": round(dt, 4),
                                 "ms_per_step_per_job": round(dt / args.steps * 1e3, 4),
                                 "aggregate_residues_per_s": round(basep * args.steps / dt, 1), "jobs_per_s": round(K * args.steps / dt, 2)})
            print(json.dumps(out["lines"][-1]), file=sys.stderr, flush=True)
    base = out["lines"][0]["aggregate_residues_per_s"]
    for l in out["lines"]:
        l["vs_one_job"] = round(l["aggregate_residues_per_s"] / base, 3)
    print(json.dumps(out))
    for j in jobs:
        j["ctx"].close()


if __name__ == "__main__":
    main()
