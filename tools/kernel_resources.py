"""Per-kernel register / LDS / occupancy table of the HIP sources (hipcc -Rpass-analysis=kernel-resource-usage; cross-compiles
without a GPU):   python tools/kernel_resources.py [file.hip ...] > profiles/rNN_kernel_resources.txt"""
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "rstnet_amd", "csrc")
KEYS = ("VGPRs", "AGPRs", "TotalSGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "SGPRs Spill", "VGPRs Spill", "LDS Size [bytes/block]")


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    return [re.sub(r"\((?!anonymous).*$", "", re.sub(r"^void ", "", n.replace("(anonymous namespace)::", ""))) for n in out]


def main(files):
    files = files or sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    print(f"{'kernel':86s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'scr':>4s} {'occ':>4s} {'sSpl':>5s} {'vSpl':>5s} {'LDS(static)':>11s}")
    for f in files:
        r = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only",
                            "-Rpass-analysis=kernel-resource-usage", "-c", f, "-o", os.devnull], capture_output=True, text=True, cwd=CSRC)
        rows, cur = [], None
        for line in r.stderr.splitlines():
            m = re.search(r"remark: Function Name: (\S+)", line)
            if m:
                cur = {"name": m.group(1)}
                rows.append(cur)
                continue
            m = re.search(r"remark:\s+(.*?): (\d+)", line)
            if m and cur is not None and m.group(1) in KEYS:
                cur[m.group(1)] = int(m.group(2))
        print(f"# {os.path.basename(f)}")
        for row, name in zip(rows, demangle([r_["name"] for r_ in rows])):
            print(f"{name[:86]:86s} " + " ".join(f"{row.get(k, 0):>{w}d}" for k, w in zip(KEYS, (5, 5, 5, 4, 4, 5, 5, 11))))


if __name__ == "__main__":
    main(sys.argv[1:])
