"""Compile the C part of the oracle (test infrastructure) into oracle/_build/."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
OUT_DIR = os.path.join(HERE, "_build")
LIB = os.path.join(OUT_DIR, "librvq_ref.so")


def build(force: bool = False) -> str:
    src = os.path.join(HERE, "rvq_ref.c")
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= os.path.getmtime(src):
        return LIB
    os.makedirs(OUT_DIR, exist_ok=True)
    cmd = ["gcc", "-O3", "-mavx2", "-mfma", "-fopenmp", "-ffp-contract=off", "-fno-math-errno", "-shared", "-fPIC",
           src, "-o", LIB, "-lm"]
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force=True))
