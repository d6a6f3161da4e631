"""Per-kernel totals of one `rocprofv3 --pmc <counters> --kernel-trace --output-format csv -- python3 bench.py ...` pass.

    python tools/pmc_bench_summary.py <rocprofv3 output dir> <summary.json> [label]

Dispatches are grouped by (kernel name, grid size) so that the M = 12288 (three CFG branches) and M = 4096
(cond-only steps) launches of one kernel stay apart.  For every group: launches, average duration from the
kernel trace (serialised by the profiler, so slower than un-profiled), and the sum / per-launch average of every
counter collected in the pass.  FETCH_SIZE and WRITE_SIZE are reported by rocprofv3 in KiB.
"""
import csv
import glob
import json
import sys
from collections import defaultdict

csv.field_size_limit(1 << 30)


def short(name: str) -> str:
    name = name.replace("void ", "")
    return name if len(name) <= 110 else name[:110]


def main(root, out_path, label=""):
    groups = defaultdict(lambda: {"launches": 0, "dur_ns": 0.0, "counters": defaultdict(float)})
    seen_dispatch = set()
    for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            key = (short(row["Kernel_Name"]), int(row["Grid_Size"]))
            g = groups[key]
            g["counters"][row["Counter_Name"]] += float(row["Counter_Value"])
            did = (f, row["Dispatch_Id"])
            if did not in seen_dispatch:
                seen_dispatch.add(did)
                g["launches"] += 1
                if "Start_Timestamp" in row and row.get("End_Timestamp"):
                    g["dur_ns"] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
    rows = []
    for (name, grid), g in groups.items():
        n = g["launches"]
        rows.append({"kernel": name, "grid_size": grid, "launches": n, "avg_duration_us": g["dur_ns"] / n / 1e3 if n else None,
                     "total_duration_ms": g["dur_ns"] / 1e6,
                     "counters_per_launch": {c: v / n for c, v in g["counters"].items()},
                     "counters_total": dict(g["counters"])})
    rows.sort(key=lambda r: -r["total_duration_ms"])
    json.dump({"label": label, "groups": rows}, open(out_path, "w"), indent=1)
    for r in rows[:14]:
        print(f'{r["kernel"][:70]:70s} grid {r["grid_size"]:>9d} n={r["launches"]:>6d} avg {r["avg_duration_us"]:.1f} us  '
              + "  ".join(f"{c}={v:.4g}" for c, v in r["counters_per_launch"].items()))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "")
