#!/usr/bin/env python3
"""Generate the 28-domain Viterbi known-answer fixture from the reference's own data files.

Run in the build container only (needs /root/reference). Reads DATA files only:
  cli/src/scer_fg_28.fasta   headers "NAME [start-end]" = the 28 PrD segments (1-based, inclusive)
  cli/example/Scer.fasta     the proteome those segments were cut from
  cli/src/prd_freq_scer_04.txt, cli/example/four_classic_prions.fasta (copied as input fixtures)
Writes (committed):
  tests/golden/kat28.tsv     gene, orf, start, end   (expected Viterbi PrD run with fg = prd_freq_scer_04)
  tests/golden/kat28.fasta   the 28 full-length proteins (input)
  tests/golden/prd_freq_scer_04.txt, tests/golden/four_classic_prions.fasta
"""
import os
import re
import shutil

REF = "/root/reference/cli"
OUT = os.path.dirname(os.path.abspath(__file__))


def read_fasta(path):
    recs, name, buf = [], None, []
    with open(path) as f:
        for line in f:
            line = line.rstrip("\r\n")
            if line.startswith(">"):
                if name is not None:
                    recs.append((name, "".join(buf)))
                name, buf = line[1:].strip(), []
            elif name is not None:
                buf.append(line)
    if name is not None:
        recs.append((name, "".join(buf)))
    return recs


def main():
    doms = read_fasta(os.path.join(REF, "src/scer_fg_28.fasta"))
    prot = read_fasta(os.path.join(REF, "example/Scer.fasta"))
    rows, subset = [], []
    for hdr, dseq in doms:
        m = re.match(r"(\S+)\s+\[(\d+)-(\d+)\]", hdr)
        gene, start, end = m.group(1), int(m.group(2)), int(m.group(3))
        hits = [(n, s) for n, s in prot if s.rstrip("*")[start - 1:end] == dseq]
        assert len(hits) == 1, (gene, len(hits))
        orf, seq = hits[0]
        rows.append((gene, orf, start, end))
        subset.append((orf, seq))
    with open(os.path.join(OUT, "kat28.tsv"), "w") as f:
        f.write("gene\torf\tstart\tend\n")
        for r in rows:
            f.write("%s\t%s\t%d\t%d\n" % r)
    with open(os.path.join(OUT, "kat28.fasta"), "w") as f:
        for orf, seq in subset:
            f.write(">%s\n%s\n" % (orf, seq))
    shutil.copy(os.path.join(REF, "src/prd_freq_scer_04.txt"), os.path.join(OUT, "prd_freq_scer_04.txt"))
    shutil.copy(os.path.join(REF, "example/four_classic_prions.fasta"), os.path.join(OUT, "four_classic_prions.fasta"))
    print("wrote", len(rows), "KAT rows")


if __name__ == "__main__":
    main()
