"""ctypes binding of the TEST-ONLY 1-lane emulation of the device logic (tests/emu/kq_emu.cpp)."""
import ctypes as C
import os
import subprocess

import numpy as np

from kueue_amd import _ffi as F
from kueue_amd.api import Decisions

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libkq_emu.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        subprocess.check_call(["make", "-C", HERE, "-s"])
        _lib = C.CDLL(LIB)
        _lib.kqe_last_error.restype = C.c_char_p
    return _lib


class EmuEngine:
    def __init__(self, cfg):
        self.h = C.c_void_p()
        assert lib().kqe_engine_create(C.byref(cfg), C.byref(self.h)) == 0

    def derive(self):
        assert lib().kqe_snapshot_derive(self.h) == 0
        n = self.snap.N * self.snap.n_fr
        sq, us, fl = np.zeros(n, np.int64), np.zeros(n, np.int64), np.zeros(n, np.uint8)
        assert lib().kqe_read_planes(self.h, F.ptr(sq), F.ptr(us), F.ptr(fl)) == 0
        return sq, us, fl

    def force_exact_drs(self, on=True):
        lib().kqe_force_exact_drs(self.h, 1 if on else 0)

    def close(self):
        if self.h:
            lib().kqe_engine_destroy(self.h)
            self.h = None

    def put(self, snap):
        rc = lib().kqe_snapshot_put(self.h, C.byref(snap.struct()))
        assert rc == 0, (rc, lib().kqe_last_error(self.h))
        self.snap = snap

    def run(self, heads, want_usage=False, tgt_cap=None):
        d = Decisions(heads, tgt_cap=tgt_cap)
        rc = lib().kqe_cycle_run(self.h, C.byref(heads.struct()), C.byref(d.struct()))
        d.rc = rc
        d.error = lib().kqe_last_error(self.h).decode()
        if want_usage and rc == 0:
            u = np.zeros(self.snap.N * self.snap.n_fr, np.int64)
            lib().kqe_read_usage(self.h, F.ptr(u))
            d.usage_after = u
        b = C.c_int64()
        lib().kqe_last_bytes(self.h, C.byref(b))
        d.bytes = b.value
        pb = np.zeros(2, np.int64)
        lib().kqe_phase_bytes(self.h, F.ptr(pb))
        d.phase_bytes = pb.tolist()
        return d
