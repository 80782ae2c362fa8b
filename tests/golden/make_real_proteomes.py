#!/usr/bin/env python3
"""Real-proteome fixtures: the reference's own INPUT data files + anchors regenerated from this repo's oracle.

Run in the build container only (needs /root/reference). Copies DATA files only (FASTA text, no source):
  cli/example/Scer.fasta          -> tests/golden/Scer.fasta          (BASELINE config 2: 5,880 yeast proteins; the
                                     input of cli/example/generate_scer_spreadsheet.sh:14-21, `-c 60 -a 1`)
  web/files/TAIR10_pep_20101214   -> tests/golden/TAIR10_pep_20101214 (59 Arabidopsis records, the web app's sample)
and writes tests/golden/real_proteome_anchors.json: figures of the ORACLE's rows on them (SURVEY.md 8(c) C4 asks for
these to be regenerated from the build's own oracle; they are NOT reference output): number of proteins with a core at
c = 60, Viterbi PrD runs / residues, the top COREscore record, SHA-256 of the oracle's row bytes for the default run and
for the two-pass `-a 0.5` run (background counted from the input itself, plaac.java:377-384, :444-500).
"""
import hashlib
import json
import os
import shutil
import sys

import numpy as np

OUT = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(OUT))
sys.path.insert(0, ROOT)
REF = "/root/reference"
FILES = {"Scer.fasta": "cli/example/Scer.fasta", "TAIR10_pep_20101214": "web/files/TAIR10_pep_20101214"}


def anchors(path):
    from oracle import oracle_ctypes as oc
    from plaac_amd import hostio
    names, codes, offs = hostio.read_fasta(path)
    out = {"records": len(names), "residues_untrimmed": int(offs[-1])}
    counts = oc.histogram(codes, offs)
    out["bg_counts"] = [int(c) for c in counts]
    for tag, kw in (("default", {}), ("alpha0.5_two_pass", dict(alpha=0.5, bgcounts=counts.astype(np.float64)))):
        rows, tr = oc.score_batch(oc.build_params(**kw), codes, offs, tracks=True, nthreads=8)
        scored = rows["prot_len"] > 0
        has_core = rows["core_start"] >= 0
        nruns = nres = 0
        for i in range(len(names)):
            v = tr["vit"][int(offs[i]):int(offs[i]) + int(rows["prot_len"][i])]
            d = np.diff(np.concatenate(([0], v, [0])).astype(np.int8))
            nruns += int((d == 1).sum())
            nres += int(v.sum())
        cs = np.where(has_core, rows["core_score"], -np.inf)
        top = int(np.argmax(cs))
        out[tag] = {
            "scored_records": int(scored.sum()), "scored_residues": int(rows["prot_len"].sum()),
            "proteins_with_core": int(has_core.sum()), "prd_runs": nruns, "prd_residues": nres,
            "top_core": None if not has_core.any() else {"seqid": names[top].decode(), "start1": int(rows["core_start"][top]) + 1,
                         "end1": int(rows["core_end"][top]) + 1, "score_bits": "%016x" % int(
                             np.float64(rows["core_score"][top]).view(np.uint64)),
                         "score": float(rows["core_score"][top])},
            "papa_centres": int((rows["papa_cen"] >= 0).sum()),
            "rows_sha256": hashlib.sha256(rows.tobytes()).hexdigest(),
        }
    return out


def main():
    res = {}
    for dst, src in FILES.items():
        shutil.copyfile(os.path.join(REF, src), os.path.join(OUT, dst))
        os.chmod(os.path.join(OUT, dst), 0o644)
        res[dst] = anchors(os.path.join(OUT, dst))
    with open(os.path.join(OUT, "real_proteome_anchors.json"), "w") as fh:
        json.dump(res, fh, indent=1, sort_keys=True)
        fh.write("\n")
    print(json.dumps(res, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
