"""TEST INFRASTRUCTURE ONLY: builds (if stale) and loads the host-emulated kernel library.

The emulated build compiles the unmodified sources in iplan_amd/csrc/ against
tests/emu/shim/hip/hip_runtime.h.  Only tests import this module; nothing under iplan_amd/ does.
"""
import ctypes
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
EMU_SO = os.path.join(ROOT, "tests", "emu", "build", "libiplan_emu.so")
_lib = None


def get_emu_lib():
    global _lib
    if _lib is None:
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "iplan_amd", "csrc"), "emu"], check=True)
        from iplan_amd._lib import Lib
        _lib = Lib(ctypes.CDLL(EMU_SO))
    return _lib
