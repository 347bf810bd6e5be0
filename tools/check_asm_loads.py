"""Static check of csrc/lm_temporal.hip's hand-managed weight stream (run on the compiler's ISA; no GPU needed):

    python tools/check_asm_loads.py            # compiles rstnet_amd/csrc/lm_temporal.hip to ISA and checks every kernel instance

The weight waves load into FIXED registers v128 .. v255 through inline asm and take every 16-byte piece out of them with one asm
statement that first waits for it; "pin" variables defined in exactly those registers keep the compiler out of them while a block is
in flight.  The checker makes sure of exactly that, on the weight waves' code (from the first asm load to the final drain): (1) no
compiler-generated instruction mentions v128 .. v255, (2) every asm load targets that range, (3) the kernel spills no VGPR and uses no
scratch, and the compiler issues no vector-memory LOAD of its own there (it would wait for it with a count that ignores the asm loads), (4) a piece is only read by a statement that starts with its own s_waitcnt, and nothing inside an asm block writes the range
but the loads."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "rstnet_amd", "csrc", "lm_temporal.hip")
REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")
FIRST = 128


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def check_kernel(lines):
    # the weight waves' code: from the first asm load to the last asm `s_waitcnt vmcnt(0)` (the drain behind the loop)
    first = next((i for i, ln in enumerate(lines) if ln.strip().startswith("global_load_dwordx4 v[1") or ln.strip().startswith("global_load_dwordx4 v[2")), None)
    last = max((i for i, ln in enumerate(lines) if ln.strip().startswith("s_waitcnt vmcnt(0)") and i > 0 and lines[i - 1].strip().startswith(";;#ASMSTART")), default=None)
    if first is None or last is None:
        return 0, 0, ["no asm loads / no drain found"]
    while not lines[first].strip().startswith(";;#ASMSTART"):
        first -= 1
    lines = lines[first:last + 2]
    bad, n_loads, n_takes, in_asm, waited = [], 0, 0, False, False
    for ln in lines:
        t = ln.strip()
        if t.startswith(";;#ASMSTART"):
            in_asm, waited = True, False
            continue
        if t.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
            continue
        t = t.split(";")[0].strip()
        hi = {r for r in regs_of(t) if r >= FIRST}
        if not in_asm:
            if hi:
                bad.append(f"compiler-generated `{t}` touches the stream's registers")
            if re.match(r"(global|flat|buffer|scratch)_load", t):
                # the compiler waits for its own loads with counts that ignore the stream's asm loads: such a wait can be satisfied early
                bad.append(f"compiler-generated vector-memory load `{t}` inside the weight waves' code")
            continue
        if t.startswith("s_waitcnt"):
            waited = True
        elif t.startswith("global_load_dwordx4"):
            n_loads += 1
            if not regs_of(t.split(",")[0]) or min(regs_of(t.split(",")[0])) < FIRST:
                bad.append(f"asm load `{t}` outside the stream's registers")
        elif hi:
            n_takes += 1
            if not waited:
                bad.append(f"`{t}` reads the stream's registers without a wait in its statement")
            if regs_of(t.split(",")[0]) & hi:
                bad.append(f"`{t}` writes the stream's registers")
    return n_loads, n_takes, bad


def main():
    isa = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-S", SRC, "-o", "-",
                          "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True, cwd=os.path.dirname(SRC))
    if isa.returncode:
        sys.exit(isa.stderr)
    cur, fail, seen = None, 0, 0
    for ln in isa.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", ln)
        if m:
            cur = m.group(1)
        if not (cur and "temporal_frame_kernel" in cur):
            continue
        m = re.search(r"remark:\s+(ScratchSize \[bytes/lane\]|VGPRs Spill|VGPRs): (\d+)", ln)
        if m:
            v = int(m.group(2))
            if m.group(1) == "VGPRs":
                seen += 1
                if v != 256:
                    print(f"FAIL {cur}: owns {v} VGPRs, the stream needs v128 .. v255"); fail += 1
            elif v:
                print(f"FAIL {cur}: {m.group(1)} = {v}"); fail += 1
    text = isa.stdout.splitlines()
    starts = [i for i, ln in enumerate(text) if re.match(r"^_ZN\S*temporal_frame_kernel\S*:", ln)]
    for s in starts:
        e = next(i for i in range(s, len(text)) if text[i].strip().startswith("s_endpgm"))
        n_loads, n_takes, bad = check_kernel(text[s + 1:e + 1])
        kname = text[s].split(":")[0]
        print(f"{kname}: {n_loads} asm loads, {n_takes} reads of the stream's registers, {len(bad)} violations")
        for b in bad[:8]:
            print("   " + b)
        fail += len(bad) + (n_loads == 0)
    if not starts or seen != len(starts):
        print(f"FAIL: {len(starts)} kernel instances in the ISA, {seen} in the resource remarks"); fail += 1
    sys.exit(1 if fail else 0)


if __name__ == "__main__":
    main()
