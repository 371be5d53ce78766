"""ctypes loader of tests/host_walk (TEST INFRASTRUCTURE ONLY): the product's walk code (csrc/walk_device.h) compiled for the host."""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent / "host_walk"
SO = HERE / "_build" / "liblrsc_host_walk.so"


class HwParams(C.Structure):
    _fields_ = [("idmer_len", C.c_int32), ("min_kmer_len", C.c_int32), ("max_leaves", C.c_int32), ("pb_coverage", C.c_int32),
                ("error_rate", C.c_double)]


def build():
    r = subprocess.run(["make", "-C", str(HERE)], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"building tests/host_walk failed:\n{r.stdout[-3000:]}\n{r.stderr[-3000:]}")


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


_CODE = np.full(256, 255, dtype=np.uint8)
for _i, _ch in enumerate(b"ACGT"):
    _CODE[_ch] = _i


class HostWalk:
    def __init__(self):
        build()
        self.lib = L = C.CDLL(str(SO))
        L.hw_index_create.restype = C.c_void_p
        L.hw_index_create.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_uint64, C.c_int, C.c_void_p, C.c_int]
        L.hw_index_free.argtypes = [C.c_void_p]
        L.hw_extend_walk.restype = C.c_int
        L.hw_extend_walk.argtypes = [C.c_void_p, C.POINTER(HwParams), C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int32, C.c_uint32,
                                     C.c_uint32, C.c_int, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32),
                                     C.POINTER(C.c_uint32)]

    def index(self, bwt_units, rbwt_units, num_symbols, wide=False, tables=()):
        a = np.ascontiguousarray(bwt_units, dtype=np.uint8)
        b = np.ascontiguousarray(rbwt_units, dtype=np.uint8)
        ks = np.ascontiguousarray(list(tables), dtype=np.int32)
        h = self.lib.hw_index_create(_p(a), a.size, _p(b), b.size, num_symbols, int(wide), _p(ks), ks.size)
        assert h, "hw_index_create failed"
        return h

    def index_free(self, h):
        self.lib.hw_index_free(h)

    def extend_walk(self, h, params, src, path, trg, dis, initk, max_overlap, min_sa, mode):
        """-> (code, mergedSeq, steps, single-leaf fast steps); mode 0 = Walk::run, 1 = wp_extend_kernel's loop."""
        hp = HwParams(params.idmer_len, params.min_kmer_len, params.max_leaves, params.pb_coverage, params.error_rate)
        q = (src[len(src) - initk:] + path + trg).encode()
        codes = _CODE[np.frombuffer(q, dtype=np.uint8)]
        assert codes.max(initial=0) < 4
        cap = 2 * len(q) + 4096
        out = np.zeros(cap, dtype=np.uint8)
        n, steps, fast = C.c_uint32(), C.c_uint32(), C.c_uint32()
        code = self.lib.hw_extend_walk(h, C.byref(hp), _p(codes), initk, len(path), len(trg), dis, max_overlap, min_sa, mode, _p(out), cap,
                                       C.byref(n), C.byref(steps), C.byref(fast))
        merged = bytes(np.frombuffer(b"ACGT", dtype=np.uint8)[out[: n.value]]).decode()
        return code, merged, steps.value, fast.value
