"""ctypes loader of tests/host_emul (TEST INFRASTRUCTURE ONLY): the correction kernels' per-lane state machine
(longreadselfcorrect_amd/csrc/walk_sm.h) driven lane by lane on the CPU."""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent / "host_emul"
SO = HERE / "_build" / "liblrsc_emul.so"

DP_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint8), C.c_uint32, C.c_uint32, C.c_uint32, C.c_double, C.c_int32,
                    C.POINTER(C.c_uint32), C.POINTER(C.c_uint8), C.c_uint32)


def build():
    r = subprocess.run(["make", "-C", str(HERE)], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"building tests/host_emul failed:\n{r.stdout[-3000:]}\n{r.stderr[-3000:]}")


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class Emul:
    def __init__(self):
        build()
        self.lib = L = C.CDLL(str(SO))
        L.emul_index_create.restype = C.c_void_p
        L.emul_index_create.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_uint64, C.c_int, C.c_void_p, C.c_int]
        L.emul_index_free.argtypes = [C.c_void_p]
        L.emul_correct_reads.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, DP_CB, C.c_void_p,
                                         C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64,
                                         C.c_void_p, C.c_void_p]

    def index(self, bwt_units, rbwt_units, num_symbols, wide=False, tables=()):
        a = np.ascontiguousarray(bwt_units, dtype=np.uint8)
        b = np.ascontiguousarray(rbwt_units, dtype=np.uint8)
        ks = np.ascontiguousarray(list(tables), dtype=np.int32)
        h = self.lib.emul_index_create(_p(a), a.size, _p(b), b.size, num_symbols, int(wide), _p(ks), ks.size)
        assert h, "emul_index_create failed"
        return h

    def index_free(self, h):
        self.lib.emul_index_free(h)

    def correct_reads(self, h, params, bases, off, seed_count, seeds, dp=None, max_walks=0, max_steps=2000):
        """-> (counters int64[n, 11], pieces list[list[str]] (one list per piece), stats (sweeps, requests, launches)).
        dp(query str, k, min_overlap, min_identity, min_call_coverage) -> (rows, consensus str) answers correctByMSAlignment."""
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        off = np.ascontiguousarray(off, dtype=np.uint64)
        n = off.size - 1
        seed_count = np.ascontiguousarray(seed_count, dtype=np.uint32)
        seeds = np.ascontiguousarray(seeds, dtype=np.int32).reshape(-1, 8)
        code_of = np.full(256, 255, dtype=np.uint8)
        for i, ch in enumerate(b"ACGT"):
            code_of[ch] = i
        codes = code_of[bases]
        counters = np.zeros((n, 11), dtype=np.int64)
        out_cap = int(off[-1]) * 4 + 4096
        out = np.zeros(out_cap, dtype=np.uint8)
        pcap = int(seed_count.sum()) + n + 2
        poff = np.zeros(pcap, dtype=np.uint64)
        npieces = C.c_uint64()
        stats = np.zeros(3, dtype=np.uint64)

        def cb(user, q, lq, k, mo, mi, mc, rows_p, cons_p, cap):
            query = "".join("ACGT"[q[i]] for i in range(lq))
            rows, cons = dp(query, k, mo, mi, mc)
            rows_p[0] = rows
            if len(cons) > cap:
                return -1
            for i, ch in enumerate(cons):
                cons_p[i] = "ACGT".index(ch)
            return len(cons)

        cbf = DP_CB(cb) if dp is not None else C.cast(None, DP_CB)
        st = self.lib.emul_correct_reads(h, C.byref(params), _p(codes), _p(off), n, _p(seed_count), _p(seeds), cbf, None, max_walks, max_steps,
                                         _p(counters), None, _p(out), out_cap, _p(poff), pcap, C.byref(npieces), _p(stats))
        if st != 0:
            raise RuntimeError(f"emul_correct_reads: status {st}")
        text = bytes(np.frombuffer(b"ACGT", dtype=np.uint8)[out[: int(poff[npieces.value])]]).decode()
        pieces = [text[int(poff[i]): int(poff[i + 1])] for i in range(npieces.value)]
        return counters, pieces, tuple(int(x) for x in stats)
