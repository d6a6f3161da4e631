"""Per-kernel totals of one `rocprofv3 --pmc <counters> --kernel-trace --output-format csv -- python3 bench.py ...` pass.

    python tools/pmc_bench_summary.py <rocprofv3 output dir> <summary.json> [label]

Dispatches are grouped by (kernel name, grid size, size class).  Persistent kernels launch one workgroup per CU
whatever the problem size, so the grid does not tell the M = 12288 launches (three CFG branches) from the M = 4096
ones (cond-only steps): launches of one (kernel, grid) whose duration is below half of that pair's longest launch
form the class "small", the others "large".  For every group: launches, average duration from the
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
    dispatches = {}
    for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            d = dispatches.setdefault((f, row["Dispatch_Id"]), {
                "key": (short(row["Kernel_Name"]), int(row["Grid_Size"])),
                "dur": float(row["End_Timestamp"]) - float(row["Start_Timestamp"]), "counters": {}})
            d["counters"][row["Counter_Name"]] = d["counters"].get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
    longest = defaultdict(float)
    for d in dispatches.values():
        longest[d["key"]] = max(longest[d["key"]], d["dur"])
    groups = defaultdict(lambda: {"launches": 0, "dur_ns": 0.0, "counters": defaultdict(float)})
    for d in dispatches.values():
        cls = "small" if d["dur"] < 0.5 * longest[d["key"]] else "large"
        g = groups[d["key"] + (cls,)]
        g["launches"] += 1
        g["dur_ns"] += d["dur"]
        for c, v in d["counters"].items():
            g["counters"][c] += v
    rows = []
    for (name, grid, cls), g in groups.items():
        n = g["launches"]
        rows.append({"kernel": name, "grid_size": grid, "size_class": cls, "launches": n, "avg_duration_us": g["dur_ns"] / n / 1e3 if n else None,
                     "total_duration_ms": g["dur_ns"] / 1e6,
                     "counters_per_launch": {c: v / n for c, v in g["counters"].items()},
                     "counters_total": dict(g["counters"])})
    rows.sort(key=lambda r: -r["total_duration_ms"])
    json.dump({"label": label, "groups": rows}, open(out_path, "w"), indent=1)
    for r in rows[:14]:
        print(f'{r["kernel"][:70]:70s} grid {r["grid_size"]:>9d} {r["size_class"]:5s} n={r["launches"]:>6d} avg {r["avg_duration_us"]:.1f} us  '
              + "  ".join(f"{c}={v:.4g}" for c, v in r["counters_per_launch"].items()))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "")
