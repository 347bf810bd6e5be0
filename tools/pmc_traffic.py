"""Per-kernel HBM traffic from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE cannot share a pass on gfx950: 3 + 2 of
the 4 TCC slots, MI355X_MICROARCH.md "rocprofv3 PMC slots").

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -o f -- python bench.py ...
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -o w -- python bench.py ...
    python tools/pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> profiles/rNN_<name>_pmc_traffic.json

Units / corrections (MI355X_MICROARCH.md, "HBM"): both counters are in KILOBYTES; on gfx950 FETCH_SIZE tallies the 128-byte
fabric read requests at 64 B, i.e. reports half of the bytes of wide coalesced reads -> doubled here.  WRITE_SIZE is used as
reported (uncalibrated per the guide).  Output: per kernel name {launches, fetch_bytes_per_launch, write_bytes_per_launch,
traffic_bytes_per_launch}; bench.py attaches the entry of its dominant kernel to the `roofline.traffic` field.
"""
import csv
import json
import re
import sys
from collections import defaultdict


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*$", "", name)


def collect(path: str, counter: str):
    acc = defaultdict(lambda: [0, 0.0])
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            a = acc[short(r["Kernel_Name"])]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    return acc


def main(fetch_csv, write_csv, out_json):
    fetch, write = collect(fetch_csv, "FETCH_SIZE"), collect(write_csv, "WRITE_SIZE")
    out = {}
    for k in sorted(set(fetch) | set(write)):
        nf, f_kb = fetch.get(k, [0, 0.0])
        nw, w_kb = write.get(k, [0, 0.0])
        fb = 2.0 * 1024.0 * f_kb / max(nf, 1)
        wb = 1024.0 * w_kb / max(nw, 1)
        out[k] = {"launches": max(nf, nw), "fetch_bytes_per_launch": round(fb), "write_bytes_per_launch": round(wb),
                  "traffic_bytes_per_launch": round(fb + wb)}
    meta = {"_note": "FETCH_SIZE x2 (gfx950 half-count of coalesced reads) + WRITE_SIZE, KB -> bytes, averaged over the launches "
                     "of each kernel name in the profiled command; separate --pmc passes"}
    with open(out_json, "w") as f:
        json.dump({**meta, **out}, f, indent=1)
    for k, v in sorted(out.items(), key=lambda kv: -kv[1]["traffic_bytes_per_launch"] * kv[1]["launches"])[:12]:
        print(f"{k[:70]:70s} n={v['launches']:6d} traffic/launch={v['traffic_bytes_per_launch'] / 1e6:10.3f} MB")


if __name__ == "__main__":
    main(*sys.argv[1:4])
