"""The hand-managed weight stream of csrc/lm_temporal.hip (loads as inline asm into fixed registers v128 .. v255, waited for by hand): the
compiler must stay out of those registers and must not spill.  tools/check_asm_loads.py compiles the file to ISA (no GPU needed) and checks
every kernel instance; see the file for what exactly."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_weight_stream_registers_are_untouched_by_the_compiler():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_asm_loads.py")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count(" 0 violations") == 8, r.stdout       # bf16 / fp32 rings x head dim 64 / 128 x persistent / repair
