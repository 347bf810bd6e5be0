"""MFMA pipe utilisation and effective shader clock per kernel from one rocprofv3 PMC pass:

    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d <dir> -o m -- <cmd>
    python tools/pmc_mfma.py <dir>/.../m_counter_collection.csv [out.json]

SQ_VALU_MFMA_BUSY_CYCLES counts cycles (summed over the SIMDs that report), GRBM_GUI_ACTIVE the GPU-busy cycles of the
dispatch; effective clock = GRBM_GUI_ACTIVE / duration.  mfma_util = MFMA busy cycles / (GRBM_GUI_ACTIVE * 1024 SIMDs)
(256 CUs x 4 SIMDs; MI355X_MICROARCH.md "rocprofv3 PMC slots").
"""
import csv
import json
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    return re.sub(r"\(.*$", "", re.sub(r"^void ", "", name))


def main(path, out=None):
    per = defaultdict(lambda: defaultdict(float))
    disp = {}
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            key = (short(r["Kernel_Name"]), r["Dispatch_Id"])
            per[key][r["Counter_Name"]] += float(r["Counter_Value"])
            disp[key] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    agg = defaultdict(lambda: [0, 0.0, 0.0, 0.0, 0.0])
    for (k, _), c in per.items():
        a = agg[k]
        a[0] += 1
        a[1] += c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        a[2] += c.get("GRBM_GUI_ACTIVE", 0.0)
        a[3] += c.get("SQ_BUSY_CYCLES", 0.0)
        a[4] += disp[(k, _)]
    res = {}
    for k, (n, mfma, gui, sqb, ns) in sorted(agg.items(), key=lambda kv: -kv[1][4]):
        if gui <= 0:
            continue
        # GRBM_GUI_ACTIVE comes back summed over the 8 XCDs (the "clock" below is 8 x the shader clock), the MFMA counter
        # summed over all 1024 SIMDs: busy fraction of a SIMD's matrix pipe = mfma / ((gui / 8) * 1024)
        res[k] = {"launches": n, "total_ms": round(ns / 1e6, 3), "effective_clock_ghz": round(gui / ns, 3),
                  "shader_clock_ghz": round(gui / ns / 8, 3), "mfma_pipe_busy_frac": round(mfma / (gui / 8 * 1024), 4),
                  "mfma_busy_over_gui_x1024simd": round(mfma / (gui * 1024), 4), "mfma_busy_raw_over_gui": round(mfma / gui, 3)}
    for k, v in list(res.items())[:12]:
        print(f"{k[:60]:60s} {v}")
    if out:
        json.dump(res, open(out, "w"), indent=1)


if __name__ == "__main__":
    main(*sys.argv[1:3])
