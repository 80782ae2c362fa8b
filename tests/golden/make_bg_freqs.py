#!/usr/bin/env python3
"""Reference-WRITTEN bytes: the background files of the web app, web/bg_freqs/bg_freqs_<ORG>.txt.

Run in the build container only (needs /root/reference). Copies DATA files only: each is the output of
`java -jar plaac.jar -b <UniProt proteome>` (cli/build_background_files.py; print_aa_params, plaac.java:2665-2669: 22 lines
"%.6f # %s" of raw residue counts over the valid proteins) and is what the web app passes back in with `-B`
(web/lib/server.rb:152-155). Besides the 28 [start-end] headers they are the only bytes in the reference tree that the
reference itself produced, so they pin (tests/test_host_io.py) the -b / -B file format - reader and writer reproduce every
file byte for byte - and BASELINE config 3 then runs the way it is worded: `-B bg_freqs_HUMAN.txt -a 0.5`
(tests/test_gpu_cli.py). The proteomes they were counted from are not in the tree (downloaded 2014), so they do not pin
the histogram kernel itself.
"""
import glob
import os
import shutil

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bg_freqs")
REF = "/root/reference/web/bg_freqs"


def main():
    os.makedirs(OUT, exist_ok=True)
    for src in sorted(glob.glob(os.path.join(REF, "bg_freqs_*.txt"))):
        dst = os.path.join(OUT, os.path.basename(src))
        shutil.copyfile(src, dst)
        os.chmod(dst, 0o644)
        print(os.path.basename(src), os.path.getsize(dst), "bytes")


if __name__ == "__main__":
    main()
