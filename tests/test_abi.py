"""The C ABI (include/iplan_hip.h): the gfx950 library loads without a GPU and exports every entry point the header
declares; the ctypes mirrors of the argument structs have the sizes the loaded library was compiled with (for the HIP
library and for the host-emulated build of the same sources).  No compute calls."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "iplan_hip.h")
HIP_LIB = os.path.join(ROOT, "iplan_amd", "libiplan_hip.so")


def declared_entry_points():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"^\s*(?:const\s+char\s*\*|int|int64_t|size_t)\s+(iplan_\w+)\s*\(", text, flags=re.M)
    assert len(names) >= 22, names
    return sorted(set(names))


@pytest.fixture(scope="module")
def hip_lib():
    if not os.path.exists(HIP_LIB):
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "iplan_amd", "csrc"), "all"], check=True, timeout=1800)
    return C.CDLL(HIP_LIB)


def test_hip_library_exports_every_declared_entry_point(hip_lib):
    missing = [n for n in declared_entry_points() if not hasattr(hip_lib, n)]
    assert not missing, missing
    hip_lib.iplan_version.restype = C.c_int
    assert hip_lib.iplan_version() >= 100


def _check_sizes(cdll):
    from iplan_amd import _lib as L
    cdll.iplan_sizeof.restype = C.c_size_t
    cdll.iplan_sizeof.argtypes = [C.c_char_p]
    structs = set(re.findall(r"^\}\s*(Iplan\w+)\s*;", open(HEADER).read(), flags=re.M))
    assert structs == set(L.STRUCT_MIRRORS), structs ^ set(L.STRUCT_MIRRORS)
    for name, mirror in L.STRUCT_MIRRORS.items():
        assert cdll.iplan_sizeof(name.encode()) == C.sizeof(mirror), (name, cdll.iplan_sizeof(name.encode()), C.sizeof(mirror))
    assert cdll.iplan_sizeof(b"NoSuchStruct") == 0


def test_struct_mirrors_match_the_hip_library(hip_lib):
    _check_sizes(hip_lib)


def test_struct_mirrors_match_the_emulated_library():
    from tests.emu.emu_lib import get_emu_lib
    _check_sizes(get_emu_lib().c)


def test_product_path_fails_loudly_without_the_library(tmp_path):
    """No CPU fallback: with the shared library missing, the first op raises instead of computing something else."""
    import subprocess
    import sys
    code = ("import os, sys; sys.path.insert(0, %r); os.environ['IPLAN_HIP_LIB'] = %r\n"
            "from iplan_amd import _lib\n"
            "try:\n    _lib.get_lib()\nexcept _lib.IplanError as e:\n    print('RAISED', 'no CPU fallback' in str(e))\n") % (ROOT, str(tmp_path / "missing.so"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert "RAISED True" in r.stdout, r.stdout + r.stderr


def test_untracked_loads_are_waited_for():
    """the inline-asm `global_load_dwordx4 ... sc1` loads of the fused rollout launch are invisible to the compiler's
    s_waitcnt bookkeeping -- on every control-flow path of the generated gfx950 code a covering vmcnt wait precedes the first touch
    of their destination registers (scripts/check_untracked_load_waits.py cross-compiles gat.hip to assembly and walks the paths)"""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("check_untracked_load_waits", os.path.join(root, "scripts", "check_untracked_load_waits.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    import shutil
    import pytest
    if shutil.which("make") is None or shutil.which(os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")) is None:
        pytest.skip("no hipcc on this box: the assembly check needs the gfx950 cross-compiler")
    n, bad = mod.check(mod.compile_asm("gat.hip"))          # (compile flags = the Makefile's own line for gat.hip, `make -n`)
    assert n >= 16 and not bad, (n, bad[:4])
