"""gat_bwd.hip's L2-warming touches (GAT_WARM) land in a register nobody reads, whenever they land: between the first touch and the
drain behind the step loop that register must not be used by ANYTHING else.  Checks the generated gfx950 assembly:
    python scripts/dev/check_sink_register.py [gat_bwd.s]      (default: compiles iplan_amd/csrc/gat_bwd.hip)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
VREG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def vregs(text):
    s = set()
    for m in VREG.finditer(text):
        if m.group(1) is not None:
            s.add(int(m.group(1)))
        else:
            s.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return s


def compile_asm():
    out = os.path.join(tempfile.mkdtemp(prefix="iplan_asm_"), "gat_bwd.s")
    csrc = os.path.join(ROOT, "iplan_amd", "csrc")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-I" + csrc,
                    "-Wno-unused-result", "-Wno-unused-command-line-argument", "-fno-slp-vectorize", "-x", "hip", "--cuda-device-only", "-S",
                    os.path.join(csrc, "gat_bwd.hip"), "-o", out], check=True)
    return out


def check(path):
    lines = open(path).read().split("\n")
    in_asm, touches, drain = False, [], None
    for i, raw in enumerate(lines):
        t = raw.strip()
        if t.startswith(";;#ASMSTART"):
            in_asm = True
        elif t.startswith(";;#ASMEND"):
            in_asm = False
        elif in_asm and t.startswith("global_load_dword v") and "sc1" not in t:
            touches.append((i, next(iter(vregs(t.split(",")[0])))))
        elif in_asm and t.startswith("s_waitcnt vmcnt(0)") and touches and drain is None and i > touches[-1][0]:
            drain = i
    assert touches and drain, "no touches / no drain found"
    regs = {r for _, r in touches}
    assert len(regs) == 1, ("the touches use more than one register", regs)
    sink = regs.pop()
    first = touches[0][0]
    tl = {i for i, _ in touches}
    bad = []
    for i in range(first, drain):
        t = lines[i].split(";")[0].strip()
        if not t or t.startswith(".") or i in tl or t.endswith(":"):
            continue
        if sink in vregs(t):
            bad.append((i + 1, t))
    return sink, len(touches), bad


if __name__ == "__main__":
    sink, n, bad = check(sys.argv[1] if len(sys.argv) > 1 else compile_asm())
    print(f"sink register v{sink}, {n} touch instructions, {len(bad)} other uses of it before the drain")
    for b in bad[:10]:
        print("  line %d: %s" % b)
    sys.exit(1 if bad else 0)
