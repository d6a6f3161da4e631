"""Idle time between consecutive kernels of a rocprofv3 --kernel-trace CSV: total, and the largest gaps with the
kernels on either side.

    python tools/gap_report.py <..._kernel_trace.csv> [min_gap_us]
"""
import csv
import sys


def main():
    path = sys.argv[1]
    min_gap = float(sys.argv[2]) if len(sys.argv) > 2 else 50.0
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60]))
    rows.sort()
    busy = sum(e - s for s, e, _ in rows)
    span = rows[-1][1] - rows[0][0]
    gaps = []
    end = rows[0][1]
    prev = rows[0][2]
    for s, e, n in rows[1:]:
        if s > end:
            gaps.append(((s - end) / 1e3, prev, n, (s - rows[0][0]) / 1e6))
        if e > end:
            end, prev = e, n
    small = sum(g[0] for g in gaps if g[0] < min_gap)
    big = [g for g in gaps if g[0] >= min_gap]
    print(f"kernels {len(rows)}  span {span / 1e6:.1f} ms  busy {busy / 1e6:.1f} ms  idle {(span - busy) / 1e6:.1f} ms")
    print(f"gaps below {min_gap} us: {len(gaps) - len(big)} totalling {small / 1e3:.1f} ms "
          f"(mean {small / max(1, len(gaps) - len(big)):.2f} us)")
    print(f"gaps of at least {min_gap} us: {len(big)} totalling {sum(g[0] for g in big) / 1e3:.1f} ms")
    for g in sorted(big, reverse=True)[:40]:
        print(f"  {g[0]:10.1f} us at {g[3]:9.1f} ms  after {g[1]}  before {g[2]}")


if __name__ == "__main__":
    main()
