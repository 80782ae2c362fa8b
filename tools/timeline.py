#!/usr/bin/env python3
"""tools/timeline.py <kernel_trace.csv> [step] [nsteps]: the kernels of one scoring step (the step-th k_plan_lengths from the end
of a rocprofv3 --kernel-trace run) as a timeline: start / end in ms from the step's first kernel, hardware queue and
stream ids as rocprofv3 reports them."""
import csv
import re
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    starts = [i for i, r in enumerate(rows) if "k_plan_lengths" in r["Kernel_Name"]]
    if len(starts) < back:
        sys.exit("fewer than %d steps in the trace" % back)
    nsteps = int(sys.argv[3]) if len(sys.argv) > 3 else 1  # steps to print (overlapping calls: a step's kernels interleave)
    a = starts[-back]
    b = starts[-back + nsteps] if back > nsteps else len(rows)
    t0 = int(rows[a]["Start_Timestamp"])
    for r in rows[a:b]:
        name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        name = re.sub(r"^void ", "", name).split("(")[0]
        print("q%-2s s%-3s %8.3f -> %8.3f  (%7.3f ms)  %s" % (
            r["Queue_Id"], r["Stream_Id"], (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6,
            (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, name[:60]))


if __name__ == "__main__":
    main()
