"""Disassembly check for the inline-asm device-coherent loads (wave_tile.h: coh_load16_untracked -- `global_load_dwordx4 ... sc1`
between #ASMSTART / #ASMEND): the compiler's s_waitcnt bookkeeping does not see them, so the SOURCE ties every value to a tracked
load issued after it (after_load).  This script proves the result on the generated gfx950 code: on EVERY control-flow path from such
a load to the first instruction that touches one of its destination registers there is an `s_waitcnt vmcnt(N)` with N <= the number
of vector-memory LOADS issued after it on that path (loads return in order, so that wait covers it; stores may pass loads and are
not counted).  Run by tests/test_abi.py::test_untracked_loads_are_waited_for (CPU, cross-compiles gat.hip to assembly) -- ADVICE r4.

    python scripts/check_untracked_load_waits.py [file.s]       (default: compiles iplan_amd/csrc/gat.hip with the Makefile's flags)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VREG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")
LOAD = re.compile(r"^(global_load|buffer_load|flat_load|scratch_load|global_atomic\w*\s.*\bsc0\b|image_)")


def hipcc_command(src="gat.hip"):
    """the compile line the shipped object is built with: `make -n` of iplan_amd/csrc/Makefile for <src>'s object (same HIPCC, HIPFLAGS,
    per-file NOSLP and $(EXTRA)), so that the assembly walked here is the assembly that ships"""
    csrc = os.path.join(ROOT, "iplan_amd", "csrc")
    obj = os.path.join(ROOT, "build", "obj", src + ".o")
    out = subprocess.run(["make", "-n", "-B", "-C", csrc, obj], check=True, capture_output=True, text=True).stdout
    for line in out.splitlines():
        words = line.split()
        if words and words[0].endswith("hipcc") and any(w.endswith("/" + src) for w in words):
            return words
    raise RuntimeError("no hipcc line for %s in `make -n`:\n%s" % (src, out))


def compile_asm(src="gat.hip"):
    import shutil
    words = hipcc_command(src)
    if shutil.which(words[0]) is None:
        raise FileNotFoundError(words[0])
    out = os.path.join(tempfile.mkdtemp(prefix="iplan_asm_"), src + ".s")
    cmd, skip = [], False
    for w in words:                                          # -c ... -o <obj>  ->  --cuda-device-only -S -o <asm>
        if skip:
            skip = False
            continue
        if w == "-o":
            skip = True
            continue
        cmd.append("--cuda-device-only" if w == "-c" else w)
    cmd[1:1] = ["-Wno-unused-command-line-argument"]
    subprocess.run(cmd + ["-S", "-o", out], check=True)
    return out


def vregs(text):
    s = set()
    for m in VREG.finditer(text):
        if m.group(1) is not None:
            s.add(int(m.group(1)))
        else:
            s.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return s


def check(path):
    ins, labels, asm_flag = [], {}, []
    in_asm = False
    for raw in open(path):
        line = raw.split(";")[0].rstrip() if not raw.lstrip().startswith(";;#ASM") else raw.strip()
        if raw.lstrip().startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if raw.lstrip().startswith(";;#ASMEND"):
            in_asm = False
            continue
        m = re.match(r"^([.\w$]+):", line)
        if m:
            labels[m.group(1)] = len(ins)
            continue
        t = line.strip()
        if not t or t.startswith("."):
            continue
        ins.append(t)
        asm_flag.append(in_asm)
    starts = [i for i, t in enumerate(ins) if asm_flag[i] and t.startswith("global_load_dwordx4") and " sc1" in t]
    bad = []
    for s0 in starts:
        dest = vregs(ins[s0].split(",")[0])
        seen, stack = set(), [(s0 + 1, 0)]
        while stack:
            i, k = stack.pop()
            while i < len(ins):
                if (i, min(k, 64)) in seen:
                    break
                seen.add((i, min(k, 64)))
                t = ins[i]
                op = t.split()[0]
                if op == "s_waitcnt":
                    m = re.search(r"vmcnt\((\d+)\)", t)
                    if m and int(m.group(1)) <= k:
                        break                                    # covered on this path
                elif op in ("s_endpgm", "s_setpc_b64", "s_swappc_b64"):
                    break
                elif op == "s_branch":
                    i = labels[t.split()[1]]
                    continue
                elif op == "s_cbranch_execnz":                   # (a wave with no active lane has no value at stake: exec != 0 on the
                    i = labels[t.split()[1]]                     #  paths that matter, so execnz is taken and execz falls through)
                    continue
                elif op == "s_cbranch_execz":
                    pass
                elif op.startswith("s_cbranch"):
                    stack.append((labels[t.split()[1]], k))
                elif vregs(t) & dest:
                    bad.append((s0, ins[s0], i, t, k))
                    break
                if LOAD.match(op) or (op.startswith("global_load") or op.startswith("buffer_load")):
                    k += 1
                i += 1
    return len(starts), bad


if __name__ == "__main__":
    n, bad = check(sys.argv[1] if len(sys.argv) > 1 else compile_asm())
    print(f"{n} untracked coherent loads; {len(bad)} reach a use of their destination without a covering s_waitcnt vmcnt")
    for b in bad[:20]:
        print("  load #%d  %s   ->   #%d  %s   (loads issued in between: %d)" % b)
    sys.exit(1 if bad or not n else 0)
