"""ctypes binding of the CPU build of csrc/solid_pairs.h and csrc/pair_cull.h (tests/solidpairs/shim.cc) -- TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os
import subprocess

_DIR = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_DIR, "libsolidpairs.so")
        csrc = os.path.join(_DIR, "..", "..", "mujoco_mpc_amd", "csrc")
        srcs = [os.path.join(_DIR, "shim.cc"), os.path.join(csrc, "solid_pairs.h"), os.path.join(csrc, "pair_cull.h")]
        if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
            subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-I", os.path.join(_DIR, "..", "..", "include"),
                                   "-o", so, srcs[0]])
        L = C.CDLL(so)
        d, f = C.POINTER(C.c_double), C.POINTER(C.c_float)
        L.sp_thin_vs_solid.restype = C.c_double
        L.sp_thin_vs_solid.argtypes = [C.c_int, d, d, d, C.c_double, C.c_double, d, d]
        L.sp_thin_vs_solid_f32.restype = C.c_float
        L.sp_thin_vs_solid_f32.argtypes = [C.c_int, f, f, f, C.c_float, C.c_float, f, f]
        L.sp_pair_never_touches.restype = C.c_int
        L.sp_pair_never_touches.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double, C.POINTER(C.c_int), d]
        L.sp_moving_pairs.restype = C.c_int
        L.sp_moving_pairs.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.c_int]
        _LIB = L
    return _LIB
