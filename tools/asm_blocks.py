#!/usr/bin/env python3
"""Basic blocks of one kernel of build/plaac_kernels.s (make asm) with their instruction mix: asm_blocks.py <kernel substring> [min instrs]"""
import collections
import re
import sys


def blocks(path, key):
    inside, cur, name, out = False, [], None, []
    for line in open(path):
        if not inside:
            if re.match(r"^_ZN.*%s.*:\s" % re.escape(key), line) or re.match(r"^%s.*:\s" % re.escape(key), line):
                inside, name, cur = True, "entry", []
            continue
        s = line.strip()
        if s.startswith(".Lfunc_end") or s.startswith("s_endpgm") and False:
            break
        m = re.match(r"^(\.LBB\d+_\d+):", s)
        if m:
            out.append((name, cur))
            name, cur = m.group(1), []
            continue
        if not s or s.startswith(";") or s.startswith("."):
            continue
        cur.append(s.split()[0])
    out.append((name, cur))
    return out


def main():
    key = sys.argv[1]
    minn = int(sys.argv[2]) if len(sys.argv) > 2 else 150
    for name, ins in blocks("build/plaac_kernels.s", key):
        if len(ins) < minn:
            continue
        h = collections.Counter(ins)
        valu = sum(v for k, v in h.items() if k.startswith("v_"))
        f64 = sum(v for k, v in h.items() if k.startswith("v_") and "f64" in k)
        lds = sum(v for k, v in h.items() if k.startswith("ds_"))
        print("%s: %d instrs, valu %d (f64 %d), lds %d, salu %d, vmem %d" % (
            name, len(ins), valu, f64, lds, sum(v for k, v in h.items() if k.startswith("s_")),
            sum(v for k, v in h.items() if k.startswith(("global_", "buffer_", "flat_")))))
        print("   " + ", ".join("%s %d" % kv for kv in h.most_common(40)))


if __name__ == "__main__":
    main()
