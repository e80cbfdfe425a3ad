#!/usr/bin/env python
"""Print the kernels of the LAST training step of a rocprofv3 kernel trace, in launch order.

usage: python tools/step_trace.py <kernel_trace.csv> [substring filter ...]
A step ends with the fused SGD kernel (sgd_flat_kernel)."""
import csv
import re
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    pats = sys.argv[2:]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if "sgd_flat" in r["Kernel_Name"]]
    step = rows[idx[-2] + 1: idx[-1] + 1]
    t0 = int(step[0]["Start_Timestamp"])
    for r in step:
        n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        n = re.sub(r"at::native::", "", n)
        if pats and not any(p in n for p in pats):
            continue
        s = int(r["Start_Timestamp"]) - t0
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        print(f"{s / 1e6:8.2f}ms {d / 1e3:8.1f}us grid={r['Grid_Size_X']:>9} st={r['Stream_Id']} {n[:100]}")
    print("step span ms", (int(step[-1]["End_Timestamp"]) - t0) / 1e6)


if __name__ == "__main__":
    main()
