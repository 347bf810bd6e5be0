"""Writes `<prefix>.meta.json` next to a measurement file: the build id of the library the measuring process loads (rst_build_id = hash
of the sources it was built from), rst_version, the commit the snapshot was taken from (.git_head, written before gpurun by the caller;
the box has no .git) and when.  bench.py quotes a committed `*_kernel_stats.csv` only while its build id equals the loaded library's.

    python tools/profile_meta.py profiles/r06_lm_kernel_stats [more prefixes ...]
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def meta():
    from rstnet_amd import _lib
    head = "unknown"
    try:
        with open(os.path.join(ROOT, ".git_head")) as f:
            head = f.read().strip()
    except OSError:
        pass
    import hashlib
    with open(_lib.LIB_PATH, "rb") as f:
        lib_sha = hashlib.sha256(f.read()).hexdigest()
    return {"build_id": _lib.build_id(), "rst_version": int(_lib.lib().rst_version()), "git_head": head, "lib_sha256": lib_sha,
            "collected": time.strftime("%Y-%m-%dT%H:%M:%SZ", time.gmtime())}


def main():
    m = meta()
    for prefix in sys.argv[1:]:
        with open(prefix + ".meta.json", "w") as f:
            json.dump(m, f, indent=1)
            f.write("\n")
    print(json.dumps(m))


if __name__ == "__main__":
    main()
