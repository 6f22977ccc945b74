"""Timeline of the headline pipeline's steady-state periods from a rocprofv3 --kernel-trace csv: for the last periods, when the step
kernel, the generator of the next block (second stream) and the period-boundary kernel(s) start and end relative to the step
kernel's start.  python tools/period_timeline.py <kernel_trace.csv> [n_periods]"""
import csv
import sys


def main():
    rows = []
    with open(sys.argv[1]) as f:
        for r in csv.DictReader(f):
            n = r["Kernel_Name"]
            if "tda::" in n:
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n.split("(")[0].replace("void tda::", "")))
    rows.sort()
    steps = [i for i, r in enumerate(rows) if r[2].startswith("k_mh_steps<64, 8") and r[1] - r[0] > 500000]
    npd = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    for i in steps[-npd - 1:-1]:
        t0 = rows[i][0]
        nxt = [j for j in steps if j > i][0]
        print("period starting at %d:" % t0)
        for s, e, n in rows[i:nxt + 1]:
            print("   %-40s start %8.1f us  end %8.1f us  (%.1f us)" % (n[:40], (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3))


if __name__ == "__main__":
    main()
