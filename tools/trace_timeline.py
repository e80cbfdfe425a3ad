#!/usr/bin/env python
"""Per-queue busy/idle analysis of one training step from a rocprofv3 kernel trace.

usage: python tools/trace_timeline.py <kernel_trace.csv> [step_index_from_end=1]
A step is delimited by consecutive sgd_flat_kernel launches."""
import collections
import csv
import sys


def short(n):
    return n.replace("(anonymous namespace)::", "").replace("void ", "")[:60]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    back = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    idx = [i for i, r in enumerate(rows) if "sgd_flat" in r["Kernel_Name"]]
    step = rows[idx[-back - 1] + 1: idx[-back] + 1]
    t0, t1 = int(step[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in step)
    print(f"{len(step)} kernels, wall {(t1 - t0) / 1e6:.2f} ms")
    byq = collections.defaultdict(list)
    for r in step:
        byq[r["Queue_Id"]].append(r)
    for q, rs in sorted(byq.items()):
        busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs)
        gaps = [int(b["Start_Timestamp"]) - int(a["End_Timestamp"]) for a, b in zip(rs, rs[1:])]
        pos = [g for g in gaps if g > 0]
        print(f"queue {q}: {len(rs)} kernels, busy {busy / 1e6:.2f} ms, gaps {sum(pos) / 1e6:.2f} ms "
              f"(median {sorted(pos)[len(pos) // 2] / 1e3 if pos else 0:.1f} us, >20us: {sum(1 for g in pos if g > 20000)})")
    iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in step)
    busy, (cs, ce) = 0, iv[0]
    for s, e in iv[1:]:
        if s > ce:
            busy += ce - cs
            cs, ce = s, e
        else:
            ce = max(ce, e)
    busy += ce - cs
    print(f"union busy {busy / 1e6:.2f} ms, idle {(t1 - t0 - busy) / 1e6:.2f} ms")
    fam = collections.Counter()
    for r in step:
        fam[short(r["Kernel_Name"]).split("<")[0].split("(")[0]] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    for k, v in fam.most_common(16):
        print(f"  {v / 1e6:7.2f} ms  {k}")
    # biggest gaps on the main queue
    mainq = max(byq, key=lambda q: len(byq[q]))
    rs = byq[mainq]
    gl = sorted(((int(b["Start_Timestamp"]) - int(a["End_Timestamp"]), short(a["Kernel_Name"]), short(b["Kernel_Name"]))
                 for a, b in zip(rs, rs[1:])), reverse=True)[:8]
    for g, a, b in gl:
        print(f"  gap {g / 1e3:7.1f} us after {a[:40]} before {b[:40]}")


if __name__ == "__main__":
    main()
