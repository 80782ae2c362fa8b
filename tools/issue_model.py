#!/usr/bin/env python3
"""tools/issue_model.py [build/plaac_kernels.s] [profiles/r04_issue_probe.txt] -> profiles/r04_issue_classes.json

The instruction-issue roof of the path with MEASURED per-class costs ("4 cycles for every VALU and
LDS wave-instruction" was an assumption).

1. tools/issue_probe.hip (run on MI355X, output committed as profiles/r04_issue_probe.txt) gives SIMD cycles per
   wave-instruction per class at 1..4 waves per SIMD - since round 4 in TRUE cycles: a sleeping wave beside every probe
   measures the shader clock the chip holds during it (2.2 - 2.4 GHz here; round 3 multiplied wall time by the nominal
   2.4 GHz and normalised everything to v_add_f64 = 4, on the guess that the spread over the waves per SIMD was the clock).
   It is not the clock: at the measured clock v_add_u32 costs 2.6 / 3.3 / 3.7 cycles with 2 / 3 / 4 waves per SIMD and
   v_add_f64 4.7 / 5.9 / 5.5 - a SIMD does not reach its nominal one-fp64-instruction-per-4-cycles in a pure stream of
   them (8 independent chains per wave), and a cost "per class" exists only per occupancy. The model takes the mean of
   the 3- and 4-wave rows (the occupancy the scoring kernels run at):
       simple 32-bit VOP1/VOP2 (add, sub, and, or, xor, mov, cndmask by vcc, f32 add)                "fast"
       shifts, mul24, max/min, every VOP3-only op, DPP, SDWA, cndmask by an SGPR pair, 64-bit int     "other"
       every fp64 op                                                                                 "f64"
       v_cmp*, v_readlane, v_readfirstlane                                                           "slow"
   SALU instructions issue beside the VALU stream of other waves and are not charged; an LDS instruction is charged its
   ISSUE slot (4 cycles), not its 11-cycle conflict-free throughput (LDS time overlaps the VALU stream of other waves).
   Because pure streams are the WORST case for a class (mixed streams cost less per instruction than the weighted sum of
   their pure costs: v_add_f64 : v_add_u32 = 1 : 3 reads 4.3 - 4.6 against 4.2 weighted), the roof is an upper bound on the
   time the instructions need, not a lower one: a step can come in under it.
2. The `make asm` listing gives, per kernel, which share of the non-fp64 VALU instructions of its loops (basic blocks
   of >= 60 instructions) is "fast".
3. tools/pmc_summary.py combines that share with the DYNAMIC counts of the PMC passes (SQ_INSTS_VALU, SQ_INSTS_VALU_*_F64,
   SQ_INSTS_LDS per kernel and step): cycles = 4 f64 + (VALU - f64) (fast_share c_fast + (1 - fast_share) c_other) + 4 LDS,
   summed over the kernels of a step, / 1024 SIMDs / clock. bench.py prints it as roofline.issue.
"""
import collections
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAST = ("v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_mov_b32", "v_not_b32",
        "v_add_f32", "v_sub_f32", "v_add_co_u32", "v_addc_co_u32", "v_sub_co_u32", "v_subb_co_u32", "v_subrev_co_u32",
        "v_xnor_b32", "v_accvgpr")
SLOW45 = ("v_cmp", "v_readlane", "v_readfirstlane")
KERNELS = ("k_fwd_post_t", "k_bwd_pair", "k_long", "k_tracksL", "k_tracks20f", "k_refine_centres", "k_fwd_pair", "k_fwd", "k_win", "k_vit", "k_core_list", "k_pack",
           "k_finish", "k_tracks20s", "k_tracks20", "k_bwd", "k_post", "k_hist", "k_core_par", "k_core_eval", "k_core_reduce",
           "k_core_chain", "k_llr_at_centre", "k_plan_lengths", "k_plan_scatter")


def probe_costs(path):
    """true cycles per wave-instruction (launch time x the clock measured during the launch), mean of the 3- and 4-wave
    rows with 8 chains, per probe row; and the clocks seen"""
    rows, clocks = collections.OrderedDict(), []
    for line in open(path):
        m = re.match(r"(.*?)\s+chains\s+(\d+)\s+waves/SIMD (\d)\s+([\d.]+) ms\s+clock\s+(\d+) MHz\s+([\d.]+) cycles", line)
        if m:
            rows.setdefault(m.group(1).strip(), {})[(int(m.group(2)), int(m.group(3)))] = float(m.group(6))
            clocks.append(int(m.group(5)))
    out, spread = {}, {}
    for name, v in rows.items():
        c = [v[(8, w)] for w in (3, 4) if (8, w) in v]
        if c:
            out[name] = round(sum(c) / len(c), 2)
        a = [v[(8, w)] for w in (2, 3, 4) if (8, w) in v]
        if len(a) == 3:
            spread[name] = round((max(a) - min(a)) / (sum(a) / 3), 3)
    return out, {"min_MHz": min(clocks), "max_MHz": max(clocks), "mean_MHz": round(sum(clocks) / len(clocks))}, spread


def classify(op, text):
    base = op.replace("_e32", "").replace("_e64", "")
    if op.startswith("v_") and "_f64" in op:
        return "rcp64" if op.startswith(("v_rcp_f64", "v_rsq_f64", "v_sqrt_f64")) else "f64"
    if not op.startswith("v_"):
        return None
    if op.startswith(SLOW45):
        return "slow"
    if "dpp" in op or "sdwa" in op or " row_" in text or "quad_perm" in text:
        return "other"
    if base == "v_cndmask_b32":
        return "fast" if text.rstrip().endswith("vcc") else "other"
    if base.startswith(FAST):
        return "fast"
    return "other"


def kernel_mix(asm_path):
    lines = open(asm_path).read().split("\n")
    mix = {}
    i = 0
    while i < len(lines):
        l = lines[i]
        if l.startswith("_ZN") and not l.startswith("\t") and ":" in l:
            name = l.split(":")[0]
            end = next((k for k in range(i, len(lines)) if lines[k].startswith(".Lfunc_end")), None)
            if end is None:  # (a data symbol behind the last function: the 2^(j/64) table of the posteriors' exp)
                break
            key = next((k for k in KERNELS if ("%d%s" % (len(k), k)) in name), None)
            if key:
                blocks, cur = [], []
                for t in lines[i + 1:end]:
                    t = t.strip()
                    if not t or t.startswith((";", ".", "//")) and not t.startswith(".LBB"):
                        continue
                    if t.startswith(".LBB") and t.endswith(":") or re.match(r"^\.LBB\d+_\d+:", t):
                        blocks.append(cur)
                        cur = []
                    else:
                        cur.append(t)
                blocks.append(cur)
                c = collections.Counter()
                for b in blocks:
                    if len(b) < 60:
                        continue
                    for t in b:
                        cls = classify(t.split()[0], t)
                        if cls:
                            c[cls] += 1
                        elif t.startswith("ds_"):
                            c["lds"] += 1
                        elif t.startswith("s_") and not t.startswith("s_waitcnt"):
                            c["salu"] += 1
                if sum(c.values()):
                    agg = mix.setdefault(key, collections.Counter())
                    agg.update(c)
            i = end
        i += 1
    out = {}
    for k, c in mix.items():
        nonf = c["fast"] + c["other"] + c["slow"]
        out[k] = {"static_loop_instructions": dict(c),
                  "fast_share_of_non_fp64_valu": round(c["fast"] / nonf, 3) if nonf else 0.0,
                  "slow_share_of_non_fp64_valu": round(c["slow"] / nonf, 3) if nonf else 0.0}
    return out


def kernels_sha():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import pmc_summary
    return pmc_summary.kernels_sha()


def main():
    asm = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "build", "plaac_kernels.s")
    probe = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "profiles", "r04_issue_probe.txt")
    costs, clock, spread = probe_costs(probe)
    fast = [costs[k] for k in costs if k.startswith(("v_add_u32 (VOP2)", "v_mov_b32 (VOP1)", "v_and_b32", "v_add_f32 (VOP2)",
                                                      "v_cndmask vcc : v_add_u32"))]
    other = [costs[k] for k in costs if k.startswith(("v_lshlrev_b32 (VOP2)", "v_mul_u32_u24", "v_max_u32", "v_add3_u32", "v_lshl_add_u32",
                                                       "v_bfe_u32", "v_mad_i32_i24", "v_alignbit", "v_or3", "v_sad_u32", "v_perm",
                                                       "v_mov_b32 dpp", "v_add_u32 sdwa", "v_cndmask_b32_e64"))]
    slow = [costs[k] for k in costs if k.startswith(("v_cmp_gt_u32 vcc (VOPC)", "v_cmp_gt_f64 vcc (VOPC)", "v_readlane", "v_readfirstlane"))]
    f64 = [costs[k] for k in costs if k in ("v_add_f64", "v_mul_f64", "v_max_f64", "v_add_f64 |abs| modifier")]
    model = {
        "cost_cycles": {"f64": round(sum(f64) / len(f64), 2), "fast": round(sum(fast) / len(fast), 2),
                        "other": round(sum(other) / len(other), 2), "slow": round(sum(slow) / len(slow), 2),
                        "rcp64": costs.get("v_rcp_f64 (transcendental)", 15.0), "lds_issue": 4.0},
        "cost_unit": "shader cycles per wave-instruction per SIMD at the clock MEASURED during each probe, mean of 3 and 4 waves per SIMD",
        "probe_rows_true_cycles": costs,
        "probe_clock": clock,
        "spread_over_2_3_4_waves_per_simd": spread,
        "kernels": kernel_mix(asm),
        "class_shares": "static: per kernel, the instruction classes of its loop blocks (>= 60 instructions) in the `make asm` listing",
        "source": "tools/issue_model.py: profiles/r04_issue_probe.txt (MI355X; the per-class issue costs are the machine's, measured in round 4) + the `make asm` listing of this tree (kernel sources sha16 " + kernels_sha() + ")",
    }
    # (round 6: the class shares of THIS tree's listing go into a file of their own; tools/pmc_summary.py takes the newest
    #  profiles/r0*_issue_classes.json, so counters collected for this tree meet the shares of this tree)
    out = os.path.join(ROOT, "profiles", os.environ.get("ISSUE_CLASSES_OUT", "r06_issue_classes.json"))
    with open(out, "w") as fh:
        json.dump(model, fh, indent=1, sort_keys=True)
        fh.write("\n")
    print(json.dumps({k: v for k, v in model.items() if k not in ("probe_rows_true_cycles", "kernels", "spread_over_2_3_4_waves_per_simd")}, indent=1)[:3000])
    print("spread over 2/3/4 waves per SIMD:", {k: spread[k] for k in ("v_add_u32 (VOP2)", "v_add_f64", "v_lshlrev_b32 (VOP2)") if k in spread})


if __name__ == "__main__":
    main()
