"""Kernel-by-kernel timeline of ONE frame from a rocprofv3 (rocpd sqlite) kernel trace: start offset, duration and the gap
to the previous kernel's end, in dispatch order -- what a chain of short dependent launches really costs.

    python tools/frame_timeline.py <results.db> <out.csv> [marker-kernel-substring] [frame-index-from-end]

A frame = the dispatches from one occurrence of the marker kernel (default `lm_ring_begin_kernel`) to the next."""
import csv
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name if len(name) < 90 else name[:87] + "..."


def main(db, out, marker="lm_ring_begin_kernel", back=2):
    con = sqlite3.connect(db)
    cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
    want = [c for c in ("name", "start", "end", "grid_x", "workgroup_x") if c in cols]
    rows = list(con.execute(f"select {', '.join(want)} from kernels order by start"))
    idx = [i for i, r in enumerate(rows) if marker in r[0]]
    if len(idx) < back + 1:
        raise SystemExit(f"only {len(idx)} occurrences of {marker}")
    lo, hi = idx[-back - 1], idx[-back]
    frame = rows[lo:hi]
    t0 = frame[0][1]
    prev_end = t0
    busy = 0
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["i", "kernel", "grid_x", "start_us", "dur_us", "gap_us"])
        for i, r in enumerate(frame):
            name, s, e = r[0], r[1], r[2]
            gx = r[3] if len(r) > 3 else 0
            w.writerow([i, short(name), gx, round((s - t0) / 1e3, 2), round((e - s) / 1e3, 2), round((s - prev_end) / 1e3, 2)])
            busy += e - s
            prev_end = max(prev_end, e)
    span = (rows[hi][1] - t0) / 1e3
    print(f"frame of {len(frame)} dispatches: span {span:.1f} us, sum of durations {busy / 1e3:.1f} us")


if __name__ == "__main__":
    a = sys.argv[1:]
    main(a[0], a[1], a[2] if len(a) > 2 else "lm_ring_begin_kernel", int(a[3]) if len(a) > 3 else 2)
