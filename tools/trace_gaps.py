#!/usr/bin/env python
"""Idle time of the device queues from a rocprofv3 kernel trace (``*_kernel_trace.csv``).

    python tools/trace_gaps.py <kernel_trace.csv> [skip_fraction]

Per queue: busy time (union of kernel intervals), idle time inside the analysed window, and the kernels that most
often PRECEDE an idle gap (with the summed gap behind them).  Also the time during which NO queue runs anything.
The first ``skip_fraction`` (default 0.4) of the trace (warm-up, autotune) is dropped."""
import collections
import csv
import sys


def short(name: str) -> str:
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name[:70]


def main():
    path = sys.argv[1]
    skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.4
    rows = []
    with open(path) as f:
        rd = csv.DictReader(f)
        for r in rd:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "0"),
                         r.get("Stream_Id", "0"), r["Kernel_Name"]))
    rows.sort()
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    lo = t0 + (t1 - t0) * skip
    rows = [r for r in rows if r[0] >= lo]
    win = (max(r[1] for r in rows) - rows[0][0]) / 1e6
    print(f"window {win:.1f} ms, {len(rows)} kernels")
    by_q = collections.defaultdict(list)
    for r in rows:
        by_q[(r[2], r[3])].append(r)
    for q, ks in sorted(by_q.items(), key=lambda kv: -len(kv[1])):
        busy, gaps, prev_end, prev_name = 0.0, collections.Counter(), None, None
        gap_n = collections.Counter()
        hist = collections.Counter()
        for s, e, _, _, name in ks:
            if prev_end is not None and s > prev_end:
                g = (s - prev_end) / 1e3
                gaps[short(prev_name) + "  ->  " + short(name)[:40]] += g
                gap_n[short(prev_name) + "  ->  " + short(name)[:40]] += 1
                hist[min(int(g // 5) * 5, 100)] += g
            busy += (e - max(s, prev_end or s)) / 1e6 if (prev_end is None or e > prev_end) else 0.0
            if prev_end is None or e > prev_end:
                prev_end, prev_name = e, name
        span = (ks[-1][1] - ks[0][0]) / 1e6
        print(f"\nqueue {q}: {len(ks)} kernels, span {span:.1f} ms, busy {busy:.1f} ms, idle {span - busy:.1f} ms")
        print("  idle by gap length (us bucket: ms): " + ", ".join(f"{k}+: {v / 1e3:.1f}" for k, v in sorted(hist.items())))
        for k, v in gaps.most_common(12):
            print(f"  {v / 1e3:8.2f} ms in {gap_n[k]:5d} gaps   {k}")
    # time with no queue busy
    ev = sorted((r[0], r[1]) for r in rows)
    cur_end, none = ev[0][1], 0.0
    for s, e in ev[1:]:
        if s > cur_end:
            none += (s - cur_end) / 1e6
        cur_end = max(cur_end, e)
    print(f"\nall queues idle: {none:.1f} ms of {win:.1f} ms")


if __name__ == "__main__":
    main()
