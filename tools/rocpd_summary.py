"""Summarise a rocprofv3 (rocpd sqlite) kernel trace into two small CSVs that can be committed under profiles/:
per-kernel totals (what `--stats` reports) and per-(kernel, grid) aggregates so that individual layers are visible.

    python tools/rocpd_summary.py gpurun_out/prof1/r1_results.db profiles/r01_bench
"""
import csv
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name if len(name) < 120 else name[:117] + "..."


def main(db, prefix):
    con = sqlite3.connect(db)
    cur = con.cursor()
    rows = list(cur.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels "
                            "group by name order by sum(duration) desc"))
    total = sum(r[2] for r in rows)
    with open(prefix + "_kernel_stats.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "percent"])
        for n, c, t, a, mn, mx in rows:
            w.writerow([short(n), c, round(t / 1e3, 1), round(a / 1e3, 2), round(mn / 1e3, 2), round(mx / 1e3, 2), round(100 * t / total, 2)])
    rows = list(cur.execute("select name, grid_x, grid_y, grid_z, workgroup_x, lds_size, vgpr_count, accum_vgpr_count, count(*), "
                            "sum(duration), avg(duration) from kernels group by name, grid_x, grid_y, grid_z "
                            "order by sum(duration) desc"))
    with open(prefix + "_dispatch_shapes.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "grid_x", "grid_y", "grid_z", "wg_x", "lds_bytes", "vgpr", "agpr", "calls", "total_us", "avg_us", "percent"])
        for n, gx, gy, gz, wx, lds, vg, ag, c, t, a in rows:
            w.writerow([short(n), gx, gy, gz, wx, lds, vg, ag, c, round(t / 1e3, 1), round(a / 1e3, 2), round(100 * t / total, 2)])
    print(f"total kernel time {total/1e6:.3f} ms over {sum(r[8] for r in rows)} dispatches")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
