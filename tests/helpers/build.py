"""Builds the test-only helper library (tests/helpers/occupy.hip -> tests/helpers/_build/libocc.so) for gfx950."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_build", "libocc.so")


def build() -> str:
    src = os.path.join(HERE, "occupy.hip")
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    if not os.path.exists(OUT) or os.path.getmtime(OUT) < os.path.getmtime(src):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "-std=c++17", "-fPIC", "-shared", "--offload-arch=gfx950", src, "-o", OUT])
    return OUT
