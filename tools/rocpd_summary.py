"""Per-kernel summary of a rocprofv3 rocpd database (the default output format): calls, total, average, share; with
--shapes also split by grid size.

    python tools/rocpd_summary.py gpurun_out/x/prof_results.db [--shapes] [--top 25]
"""
import argparse
import re
import sqlite3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--shapes", action="store_true")
    ap.add_argument("--top", type=int, default=25)
    a = ap.parse_args()
    c = sqlite3.connect(a.db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    group = "s.kernel_name, d.grid_size_x" if a.shapes else "s.kernel_name"
    sel = "s.kernel_name, " + ("d.grid_size_x" if a.shapes else "0")
    rows = c.execute(f"select {sel}, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start) from {kd} d "
                     f"join {ks} s on d.kernel_id=s.id group by {group} order by 4 desc").fetchall()
    tot = sum(r[3] for r in rows)
    for name, grid, cnt, t, avg, mn in rows[:a.top]:
        name = re.sub(r"\(.*", "", name)[:64]
        print(f"{name:64s} grid {grid:8d} x{cnt:6d} {t / 1e6:9.2f} ms  avg {avg / 1e3:8.2f} us  min {mn / 1e3:8.2f} us {100 * t / tot:5.1f}%")
    print(f"total {tot / 1e6:.2f} ms")


if __name__ == "__main__":
    main()
