"""Merge the per-pass summaries written by tools/pmc_bench_summary.py (FETCH_SIZE, WRITE_SIZE and SQ/GRBM passes over
bench.py) into one compact per-kernel table that is committed under profiles/ and read by bench.py.

    python tools/pmc_to_profile.py <dir with pmc_fetch.json pmc_write.json pmc_sq.json> <out.json> "<command line>"

Per (kernel, size class): launches, average duration under the profiler, fabric-side bytes per launch
(2 x FETCH_SIZE + WRITE_SIZE: on gfx950 FETCH_SIZE counts 64 B per 128-B request; both counters are reported in
KiB and include Infinity-Cache hits), matrix-pipe busy fraction (SQ_VALU_MFMA_BUSY_CYCLES over 4 SIMDs x CUs x
cycles) and the clock the chip held (GRBM_GUI_ACTIVE is summed over the 8 XCDs).
"""
import json
import os
import sys

CUS, SIMDS, XCDS = 256, 4, 8


def load(path):
    if not os.path.exists(path):
        return {}
    return {(g["kernel"], g["grid_size"], g.get("size_class", "large")): g for g in json.load(open(path))["groups"]}


def main(root, out_path, command=""):
    fetch, write, sq = (load(os.path.join(root, f"pmc_{n}.json")) for n in ("fetch", "write", "sq"))
    rows = []
    for key in sorted(set(fetch) | set(write) | set(sq), key=lambda k: -(sq.get(k) or fetch.get(k) or write[k])["total_duration_ms"]):
        base = sq.get(key) or fetch.get(key) or write[key]
        row = {"kernel": key[0], "grid_size": key[1], "size_class": key[2], "launches": base["launches"],
               "avg_duration_us_profiled": base["avg_duration_us"]}
        if key in fetch:
            row["fetch_bytes_per_launch"] = 2.0 * 1024.0 * fetch[key]["counters_per_launch"]["FETCH_SIZE"]
        if key in write:
            row["write_bytes_per_launch"] = 1024.0 * write[key]["counters_per_launch"]["WRITE_SIZE"]
        if key in sq:
            c = sq[key]["counters_per_launch"]
            cycles = c["GRBM_GUI_ACTIVE"] / XCDS
            clock = cycles / (sq[key]["avg_duration_us"] * 1e3)
            # GRBM_GUI_ACTIVE / duration reads high on launches shorter than about 0.3 ms (2.6 - 3.8 "GHz" on a
            # 2.4 GHz part): below that the quotient is kept under another name and the busy fraction, which divides
            # by the same cycle count, is flagged with it
            if sq[key]["avg_duration_us"] >= 300.0:
                row["clock_ghz_held"] = clock
            else:
                row["clock_quotient_unreliable_short_launch"] = clock
            row["mfma_busy_frac"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (CUS * SIMDS * cycles)
            for name in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
                if name in c and c.get("SQ_WAVE_CYCLES"):
                    row[name.lower() + "_frac_of_wave_cycles"] = c[name] / c["SQ_WAVE_CYCLES"]
            row["lds_bank_conflict_cycles"] = c.get("SQ_LDS_BANK_CONFLICT")
        rows.append(row)
    json.dump({"command": command, "kernels": rows[:40]}, open(out_path, "w"), indent=1)
    for r in rows[:12]:
        print({k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items()})


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "")
