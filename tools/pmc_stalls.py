"""Wave-stall breakdown per kernel (and grid size, to tell layers apart) from one rocprofv3 PMC pass:

    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT \
              SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES --kernel-trace --output-format csv -d <dir> -o m -- <cmd>
    python tools/pmc_stalls.py <dir>

WAIT_ANY (parked on s_waitcnt / barrier) + WAIT_INST_ANY (issue stall: MFMA dependency / pipe busy) + ACTIVE_INST_ANY ~ WAVE_CYCLES
(MI355X_MICROARCH.md "rocprofv3 PMC slots")."""
import csv
import glob
import re
import sys
from collections import defaultdict


def main(root):
    path = glob.glob(root + "/**/*counter_collection.csv", recursive=True)[0]
    agg = defaultdict(lambda: defaultdict(float))
    seen = set()
    for r in csv.DictReader(open(path)):
        k = re.sub(r"\(.*$", "", re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]))
        key = k + " grid=" + r["Grid_Size"]
        agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
        if (key, r["Dispatch_Id"]) not in seen:
            seen.add((key, r["Dispatch_Id"]))
            agg[key]["ns"] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            agg[key]["n"] += 1
    for k, c in sorted(agg.items(), key=lambda kv: -kv[1]["ns"])[:16]:
        w = c["SQ_WAVE_CYCLES"] or 1.0
        print("%-64s n=%3d ms=%7.2f wait_any=%.2f wait_inst=%.2f (lds %.2f) active=%.2f bankconf/lds_active=%.2f waves_resident=%.2f" % (
            k[:64], c["n"], c["ns"] / 1e6, c["SQ_WAIT_ANY"] / w, c["SQ_WAIT_INST_ANY"] / w, c["SQ_WAIT_INST_LDS"] / w,
            c["SQ_ACTIVE_INST_ANY"] / w, c["SQ_LDS_BANK_CONFLICT"] / max(1.0, c["SQ_LDS_IDX_ACTIVE"]),
            c["SQ_WAVE_CYCLES"] / max(1.0, c["SQ_BUSY_CYCLES"])))      # average waves resident per busy SQ (both in quad-cycles)


if __name__ == "__main__":
    main(sys.argv[1])
