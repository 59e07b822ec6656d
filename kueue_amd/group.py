"""ctypes binding of include/kq_group.h: one root cohort tree over several GPUs of ONE process (kq_engine per device + RCCL all-reduce of
the nominations, kueue_amd/csrc/kq_group.cpp). The Go drop-in calls the same entry points through shim/go/group.go; the one-process-per-GPU
form of the same protocol is sharding.ShardedCycle over torch.distributed."""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import numpy as np

from . import _ffi as F
from .api import Decisions, Heads, Snapshot, make_config
from .engine import EngineError

HOST_COLLECTIVE = 1   # KQ_GROUP_HOST_COLLECTIVE: the exchange through pinned host memory, no RCCL; a device ordinal may repeat
FORCE_SHARDED = 2     # KQ_GROUP_FORCE_SHARDED: one device also takes the sharded path

GROUP_ABI_SYMBOLS = ["kq_group_create", "kq_group_create_opts", "kq_group_destroy", "kq_group_size", "kq_group_snapshot_put", "kq_group_cycle_run",
                     "kq_group_cycle_commit", "kq_group_cycle_release", "kq_group_read_usage", "kq_group_last_error", "kq_group_collective_info"]


class Group:
    def __init__(self, cfg: Optional[F.kq_config] = None, devices: Sequence[int] = (0,), flags: Optional[int] = None):
        """flags None: kq_group_create (KQ_GROUP_COLLECTIVE / KQ_GROUP_FORCE_SHARDED from the environment); else kq_group_create_opts."""
        self._lib = F.load_engine()
        l = self._lib
        l.kq_group_create.restype = C.c_int
        l.kq_group_last_error.restype = C.c_char_p
        l.kq_group_last_error.argtypes = [C.c_void_p]
        l.kq_group_destroy.argtypes = [C.c_void_p]
        l.kq_group_destroy.restype = None
        self.cfg = cfg if cfg is not None else make_config()
        self.devices = np.ascontiguousarray(devices, np.int32)
        self._h = C.c_void_p()
        if flags is None:
            rc = l.kq_group_create(C.byref(self.cfg), C.c_int32(len(self.devices)), F.ptr(self.devices), C.byref(self._h))
        else:
            l.kq_group_create_opts.restype = C.c_int
            rc = l.kq_group_create_opts(C.byref(self.cfg), C.c_int32(len(self.devices)), F.ptr(self.devices), C.c_uint32(flags), C.byref(self._h))
        if rc != 0:
            raise EngineError(rc, l.kq_strerror(rc).decode())
        self.snap: Optional[Snapshot] = None

    def close(self):
        if self._h:
            self._lib.kq_group_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int):
        if rc != 0:
            raise EngineError(rc, self._lib.kq_group_last_error(self._h).decode(errors="replace"))

    @property
    def size(self) -> int:
        return int(self._lib.kq_group_size(self._h))

    def put(self, snap: Snapshot):
        self._check(self._lib.kq_group_snapshot_put(self._h, C.byref(snap.struct())))
        self.snap = snap

    def run(self, heads: Heads, tgt_cap: Optional[int] = None, rsn_cap: int = 0) -> Decisions:
        d = Decisions(heads, tgt_cap=tgt_cap, rsn_cap=rsn_cap)
        self._check(self._lib.kq_group_cycle_run(self._h, C.byref(heads.struct()), C.byref(d.struct())))
        return d

    def commit(self) -> int:
        n = C.c_int32(0)
        self._check(self._lib.kq_group_cycle_commit(self._h, C.byref(n)))
        return int(n.value)

    def release(self, age: int):
        self._check(self._lib.kq_group_cycle_release(self._h, C.c_int32(age)))

    def usage(self, rank: int = 0) -> np.ndarray:
        u = np.zeros(self.snap.N * self.snap.n_fr, np.int64)
        self._check(self._lib.kq_group_read_usage(self._h, C.c_int32(rank), F.ptr(u)))
        return u

    def collective_info(self) -> dict:
        """kq_group_collective_info: what the exchange actually ran on — communicators ncclCommInitAll created, ncclAllReduce groups issued,
        exchanges summed through host memory."""
        r, a, hs = C.c_int32(0), C.c_int64(0), C.c_int64(0)
        self._check(self._lib.kq_group_collective_info(self._h, C.byref(r), C.byref(a), C.byref(hs)))
        return dict(rccl_ranks=int(r.value), allreduce_calls=int(a.value), host_sums=int(hs.value))
