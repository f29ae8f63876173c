#!/usr/bin/env python3
"""Per-step timeline from a rocprofv3 --kernel-trace CSV: busy time, idle gaps and overlap per HIP queue.

usage: python tools/timeline.py <..._kernel_trace.csv> [marker-kernel-substring]
The step boundary is the last `marker` kernel (default: sgd_kernel) — one per training step.
"""
import csv
import sys
from collections import defaultdict


def main():
    path = sys.argv[1]
    marker = sys.argv[2] if len(sys.argv) > 2 else "sgd_kernel"
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0")))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if marker in r[2]]
    if len(marks) < 3:
        print("not enough steps")
        return
    lo, hi = marks[-3] + 1, marks[-2] + 1  # one full steady-state step
    step = rows[lo:hi]
    t0, t1 = step[0][0], max(r[1] for r in step)
    print(f"step: {len(step)} launches, wall {(t1 - t0) / 1e3:.1f} us (previous marker end -> this marker end: {(rows[marks[-2]][1] - rows[marks[-3]][1]) / 1e3:.1f} us)")
    by_q = defaultdict(list)
    for r in step:
        by_q[r[3]].append(r)
    for q, rs in by_q.items():
        busy = sum(e - s for s, e, _, _ in rs)
        print(f"queue {q}: {len(rs)} launches, busy {busy / 1e3:.1f} us")
    # union busy time over all queues
    ev = sorted((s, e) for s, e, _, _ in step)
    union, cur_s, cur_e = 0, ev[0][0], ev[0][1]
    for s, e in ev[1:]:
        if s > cur_e:
            union += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    union += cur_e - cur_s
    print(f"device busy (any queue) {union / 1e3:.1f} us, idle {(t1 - t0 - union) / 1e3:.1f} us")
    print(f"{'start':>8} {'dur':>7} {'gap':>6} q  kernel")
    last_end = {}
    for s, e, name, q in step:
        gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
        last_end[q] = e
        short = name.replace("anh::(anonymous namespace)::", "").replace("void ", "")
        short = short.split("(")[0][:70]
        print(f"{(s - t0) / 1e3:8.1f} {(e - s) / 1e3:7.1f} {gap:6.1f} {q}  {short}")


if __name__ == "__main__":
    main()
