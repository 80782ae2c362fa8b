#!/usr/bin/env python3
"""Condense the rocprofv3 output of tools/pmc.sh: per-kernel mean counters, durations, HBM traffic.

FETCH_SIZE / WRITE_SIZE are in KB. MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE reports half the
bytes of a wide coalesced streaming read and other access widths are uncalibrated, so the summary carries
 (a) the raw figure, (b) the guide's x2-corrected figure (used as `traffic`), and (c) a calibration on this
repo's own access shapes (plaac_calibration_reads: three streaming reads of known size).
The instruction-issue roof (`issue_model`) combines the dynamic instruction counts of the PMC passes with the measured
per-class issue costs and the static class shares of profiles/r04_issue_classes.json (tools/issue_model.py)."""
import collections
import csv
import glob
import json
import os
import sys

KEYS = ("k_bwd_fwd_post", "k_fwd_post", "k_fwd_direct", "k_long", "k_core_par", "k_calib_read<16, 0>", "k_calib_read<4, 0>", "k_calib_read<4, 1>", "k_tracksL", "k_tracks20f", "k_refine_centres", "k_core_list", "k_core_chain", "k_core_eval", "k_core_reduce", "k_fwd_pair", "k_finish", "k_tracks20s", "k_tracks20", "k_tracks<", "k_bwd", "k_post", "k_llr_at_centre", "k_replicate", "k_vit", "k_fwd", "k_win", "k_hist", "k_plan_lengths", "k_plan_scan",
        "k_plan_scatter", "k_pack", "k_group_rows", "k_scan_u32")


def kernels_sha():
    """sha256 (first 16 hex digits) over the KERNEL sources of the tree (csrc/kernels_*.hip.inc: the device code; the host's
    scheduling and launcher files are not part of it): bench.py quotes the counters only for THIS device code"""
    import hashlib
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "plaac_amd", "csrc")
    h = hashlib.sha256()
    for f in sorted(os.listdir(root)):
        if f.startswith("kernels_") and f.endswith(".hip.inc"):
            h.update(f.encode())
            h.update(open(os.path.join(root, f), "rb").read())
    return h.hexdigest()[:16]


def kernel_key(name):
    for k in KEYS:
        if k in name:
            return k.rstrip("<")
    return None


def main(root):
    counters = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(root, "*", "*", "*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            k = kernel_key(r["Kernel_Name"])
            if k:
                counters[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    durations = {}
    for mode in ("trace_concurrent", "trace_serial"):
        d = collections.defaultdict(list)
        for f in glob.glob(os.path.join(root, mode, "*", "*_kernel_trace.csv")):
            for r in csv.DictReader(open(f)):
                k = kernel_key(r["Kernel_Name"])
                if k:
                    d[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
        durations[mode] = {k: {"calls": len(v), "mean_ms": sum(v) / len(v), "min_ms": min(v), "max_ms": max(v)}
                           for k, v in d.items()}
    bench = {}
    for mode in ("trace_concurrent", "trace_serial"):
        try:
            bench[mode] = json.loads(open(os.path.join(root, mode + ".json")).read().strip().splitlines()[-1])
        except Exception as e:  # noqa: BLE001
            bench[mode] = {"error": str(e)}
    mean = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in counters.items()}
    # launches per step (k_plan_lengths runs once per step): k_refine_centres and k_tracks20 run twice, the kernels of a
    # call cut into runs of wave-groups once per run; every figure "per step" below is the mean per launch times this
    per_step = {}
    for k, d in counters.items():
        n = [len(v) / len(counters["k_plan_lengths"][c]) for c, v in d.items() if counters.get("k_plan_lengths", {}).get(c)]
        per_step[k] = round(sum(n) / len(n), 3) if n and k != "k_hist" and not k.startswith("k_calib") else 1.0
    step = {k: {c: v * per_step[k] for c, v in d.items()} for k, d in mean.items()}
    summary = {"counters_mean_per_launch": mean, "launches_per_step": per_step, "kernel_durations": durations, "bench_lines": bench}
    cfg = (bench.get("trace_serial", {}).get("config") or bench.get("trace_concurrent", {}).get("config") or {})
    R, P = cfg.get("residues_per_gpu"), cfg.get("sequences_per_gpu")
    traffic = {}
    for k, c in step.items():
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            traffic[k] = {"fetch_raw_bytes": c["FETCH_SIZE"] * 1024, "write_bytes": c["WRITE_SIZE"] * 1024,
                          "hbm_bytes_gfx950_corrected": (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024}
    # calibration on known byte counts (plaac_calibration_reads streams R bytes in three access shapes): bytes the
    # counter reports per byte read, for 16-byte-per-lane loads and for dword loads; each scoring kernel is then
    # corrected with the factor of the loads that dominate its reads
    calib = {}
    for key, name in (("k_calib_read<16, 0>", "wide16"), ("k_calib_read<4, 0>", "dword"), ("k_calib_read<4, 1>", "dword_unaligned")):
        if R and key in traffic and traffic[key]["fetch_raw_bytes"] > 0:
            calib[name] = {"fetch_raw_bytes": traffic[key]["fetch_raw_bytes"], "bytes_read": R,
                           "bytes_per_reported_byte": R / traffic[key]["fetch_raw_bytes"]}
    summary["fetch_calibration_reads"] = calib
    NARROW = ("k_tracks20f", "k_refine_centres", "k_tracks20s", "k_tracks20", "k_tracks", "k_plan_lengths",
              "k_llr_at_centre")  # residues read as (unaligned) dwords / bytes; every other kernel reads 16 bytes per lane
    if "wide16" in calib and "dword_unaligned" in calib:
        for k, t in traffic.items():
            f = calib["dword_unaligned" if k in NARROW else "wide16"]["bytes_per_reported_byte"]
            t["fetch_factor_calibrated"] = f
            t["hbm_bytes_calibrated"] = f * t["fetch_raw_bytes"] + t["write_bytes"]
    summary["traffic"] = traffic
    # instruction-issue roof with measured per-class costs
    issue = None
    try:
        cls = json.load(open(sorted(glob.glob(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles",
                                                           "r0*_issue_classes.json")))[-1]))  # (the newest tree's class shares)
        cost = cls["cost_cycles"]
        per, tot = {}, 0.0
        for k, c in step.items():
            if k == "k_hist" or k.startswith("k_calib") or "SQ_INSTS_VALU" not in c:
                continue
            f64 = c.get("SQ_INSTS_VALU_ADD_F64", 0) + c.get("SQ_INSTS_VALU_MUL_F64", 0) + c.get("SQ_INSTS_VALU_FMA_F64", 0)
            rest = max(0.0, c["SQ_INSTS_VALU"] - f64 - c.get("SQ_INSTS_VALU_TRANS_F64", 0))
            mix = cls["kernels"].get(k, {})
            fs, ss = mix.get("fast_share_of_non_fp64_valu", 0.3), mix.get("slow_share_of_non_fp64_valu", 0.05)
            cyc = (f64 * cost["f64"] + c.get("SQ_INSTS_VALU_TRANS_F64", 0) * cost["rcp64"] +
                   rest * (fs * cost["fast"] + ss * cost["slow"] + (1 - fs - ss) * cost["other"]) +
                   c.get("SQ_INSTS_LDS", 0) * cost["lds_issue"])
            per[k] = {"cycles": round(cyc), "valu": round(c["SQ_INSTS_VALU"]), "fp64": round(f64), "lds": round(c.get("SQ_INSTS_LDS", 0)),
                      "fast_share": fs}
            tot += cyc
        issue = {"cycles_per_step": round(tot), "cost_cycles": cost, "per_kernel": per,
                 "ms_at_2.4GHz": round(tot / 1024 / 2.4e9 * 1e3, 3),
                 "what": "sum over the kernels of a step of wave-instructions x measured issue cost of their class "
                         "(tools/issue_probe.hip, tools/issue_model.py); / 1024 SIMDs / clock"}
    except (OSError, ValueError, KeyError) as e:
        issue = {"error": str(e)}
    summary["issue_model"] = issue
    json.dump(summary, open(os.path.join(root, "summary.json"), "w"), indent=1, sort_keys=True)
    # fp64 operations the kernels EXECUTED (wave instructions x 64 lanes), per residue, over one step
    f64 = sum(c.get("SQ_INSTS_VALU_ADD_F64", 0) + c.get("SQ_INSTS_VALU_MUL_F64", 0) + c.get("SQ_INSTS_VALU_FMA_F64", 0)
              for k, c in step.items() if k != "k_hist" and not k.startswith("k_calib"))
    if R:
        mode = cfg.get("mode") == "tracks"
        wl = {"cfg2": 2, "cfg3": 3, "cfg4": 4}.get(cfg.get("workload", "")[:4], 4)
        sweep = "sweep" in (cfg.get("mode") or "")
        json.dump({"workload": [wl, P, mode, sweep], "kernels_source_sha16": kernels_sha(), "source": "tools/pmc.sh (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate "
                   "passes); bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 per MI355X_MICROARCH.md gfx950 correction",
                   "bytes_per_launch": {k: round(v["hbm_bytes_gfx950_corrected"]) for k, v in traffic.items()
                                        if not k.startswith("k_calib")},
                   "bytes_per_step": round(sum(v["hbm_bytes_gfx950_corrected"] for k, v in traffic.items()
                                               if k != "k_hist" and not k.startswith("k_calib"))),
                   # the same with the read factor measured on this repo's own access shapes (fetch_calibration_reads)
                   "bytes_per_launch_calibrated": {k: round(v["hbm_bytes_calibrated"]) for k, v in traffic.items()
                                                   if "hbm_bytes_calibrated" in v and not k.startswith("k_calib")} or None,
                   "bytes_per_step_calibrated": round(sum(v.get("hbm_bytes_calibrated", 0) for k, v in traffic.items()
                                                          if k != "k_hist" and not k.startswith("k_calib"))) or None,
                   "fetch_calibration": {k: round(v["bytes_per_reported_byte"], 3) for k, v in calib.items()},
                   "fp64_ops_per_residue_executed": round(f64 * 64 / R, 1) if f64 else None,
                   "issue_model": issue,
                   # what actually bounds the path: vector-ALU and LDS wave-instructions issued per step (SQ_INSTS_VALU +
                   # SQ_INSTS_LDS over all kernels of a step); a SIMD issues one per 4 cycles at best
                   "valu_lds_wave_instructions_per_step": round(sum(
                       c.get("SQ_INSTS_VALU", 0) + c.get("SQ_INSTS_LDS", 0) for k, c in step.items()
                       if k != "k_hist" and not k.startswith("k_calib")))},
                  open(os.path.join(root, "pmc_traffic.json"), "w"), indent=1, sort_keys=True)
    for k in sorted(mean):
        t = traffic.get(k, {})
        ser = durations.get("trace_serial", {}).get(k, {})
        print("%-16s serial %.3f ms  fetch_raw %.1f MB  write %.1f MB" % (
            k, ser.get("mean_ms", float("nan")), t.get("fetch_raw_bytes", 0) / 1e6, t.get("write_bytes", 0) / 1e6))
    print("issue model:", json.dumps(issue)[:600])


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc")
