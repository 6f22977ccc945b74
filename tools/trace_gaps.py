"""Idle time between consecutive kernels of a rocprofv3 --kernel-trace csv: per kernel name the launches, busy time and the gap
that preceded each launch (queue idle).  python tools/trace_gaps.py <kernel_trace.csv> [last_fraction]"""
import csv
import sys
from collections import defaultdict


def main():
    rows = []
    with open(sys.argv[1]) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
    rows = rows[int(len(rows) * (1 - frac)):]
    busy, gap, n = defaultdict(int), defaultdict(int), defaultdict(int)
    prev_end = None
    for s, e, name in rows:
        name = name.split("(")[0][:60]
        busy[name] += e - s
        n[name] += 1
        if prev_end is not None:
            gap[name] += max(0, s - prev_end)
        prev_end = max(prev_end or e, e)
    span = rows[-1][1] - rows[0][0]
    print("window %.3f ms, %d launches, busy %.3f ms, idle %.3f ms" % (span / 1e6, len(rows), sum(busy.values()) / 1e6, sum(gap.values()) / 1e6))
    for name in sorted(busy, key=lambda k: -busy[k] - gap[k]):
        print("%-62s n=%5d busy %8.3f ms (%6.2f us each)  gap before %8.3f ms (%6.2f us each)" %
              (name, n[name], busy[name] / 1e6, busy[name] / n[name] / 1e3, gap[name] / 1e6, gap[name] / n[name] / 1e3))


if __name__ == "__main__":
    main()
