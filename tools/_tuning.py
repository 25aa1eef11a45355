"""Switches of the DEVELOPMENT build of the library (csrc: `make tuning` -> agplace_amd/lib/libagplace_hip_tuning.so): the
measured-and-rejected kernel variants and debug hooks that the release library does not contain.  The A/B harnesses under tools/
import this module FIRST (it points agplace_amd._lib at the tuning library through AGP_HIP_LIB) and set switches with
set_switch(key, value) -> agp_debug_set; the product and the tests never load that library.

    make -C agplace_amd/csrc tuning            (add EXTRA=-DAGP_CENSUS=1 for tools/census*.py)
"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "agplace_amd", "lib", "libagplace_hip_tuning.so")
if not os.path.exists(LIB):
    sys.exit(f"{LIB} is missing: run `make -C agplace_amd/csrc tuning` first")
os.environ["AGP_HIP_LIB"] = LIB
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def set_switch(key, value):
    from agplace_amd import _lib
    lib = _lib.load()
    fn = lib.agp_debug_set
    fn.restype, fn.argtypes = ctypes.c_int, [ctypes.c_char_p, ctypes.c_int]
    rc = fn(key.encode(), int(value))
    if rc != 0:
        raise RuntimeError(f"agp_debug_set({key}, {value}) failed: {rc}")
