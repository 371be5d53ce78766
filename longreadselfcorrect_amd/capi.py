"""ctypes binding of include/lrsc.h (the C ABI of liblrsc_hip.so).

Plumbing only: numpy arrays in, numpy arrays out.  Everything that computes runs in the HIP
library; if the library is missing this module raises at import (no fallback path exists).
"""
from __future__ import annotations

import ctypes as C
import os
import re
from pathlib import Path

import numpy as np

_PKG = Path(__file__).resolve().parent
_REPO = _PKG.parent


def lib_path() -> Path:
    return _PKG / "_build" / "liblrsc_hip.so"


class LrscError(RuntimeError):
    def __init__(self, status: int, what: str, detail: str):
        super().__init__(f"{what}: status {status} ({detail})")
        self.status = status
        self.detail = detail


class Interval(C.Structure):
    _fields_ = [("lower", C.c_int64), ("upper", C.c_int64)]


class BiInterval(C.Structure):
    _fields_ = [("fwd", Interval), ("rvc", Interval)]


class RankQuery(C.Structure):
    _fields_ = [("idx", C.c_int64), ("base", C.c_uint8), ("strand", C.c_uint8), ("pad", C.c_uint8 * 6)]


class IndexInfo(C.Structure):
    _fields_ = [
        ("num_strings", C.c_uint64),
        ("num_symbols", C.c_uint64),
        ("num_runs", C.c_uint64 * 2),
        ("pred_count", (C.c_uint64 * 5) * 2),
        ("block_bytes", C.c_uint32),
        ("block_symbols", C.c_uint32),
        ("device_bytes", C.c_uint64),
    ]


class Params(C.Structure):
    _fields_ = [
        ("pb_coverage", C.c_int32),
        ("error_rate", C.c_double),
        ("start_kmer_len", C.c_int32),
        ("offset", C.c_int32 * 3),
        ("mode", C.c_int32),
        ("manual", C.c_int32),
        ("scan_kmer_len", C.c_int32),
        ("kmer_len_up_bound", C.c_int32),
        ("radius", C.c_int32),
        ("hh_ratio", C.c_float),
        ("next_target", C.c_int32),
        ("max_leaves", C.c_int32),
        ("idmer_len", C.c_int32),
        ("min_kmer_len", C.c_int32),
        ("split", C.c_int32),
        ("no_dp", C.c_int32),
    ]


class WalkDesc(C.Structure):
    _fields_ = [("seq_off", C.c_uint64), ("src_len", C.c_uint32), ("path_len", C.c_uint32), ("trg_len", C.c_uint32),
                ("dis", C.c_int32), ("init_kmer", C.c_uint32), ("max_overlap", C.c_uint32), ("min_sa_threshold", C.c_uint32),
                ("pad", C.c_uint32)]


class WalkResult(C.Structure):
    _fields_ = [("code", C.c_int32), ("steps", C.c_uint32), ("out_off", C.c_uint64), ("out_len", C.c_uint32), ("pad", C.c_uint32)]


class DpJob(C.Structure):
    _fields_ = [("s1_off", C.c_uint64), ("s2_off", C.c_uint64), ("s1_len", C.c_uint32), ("s2_len", C.c_uint32),
                ("start1", C.c_int32), ("start2", C.c_int32)]


class DpResult(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("match0_start", "match0_end", "match1_start", "match1_end", "score", "edit_distance",
                                         "total_columns")] + [("cigar_len", C.c_uint32), ("cigar_off", C.c_uint64)]


class MsaQuery(C.Structure):
    _fields_ = [("seq_off", C.c_uint64), ("len", C.c_uint32), ("kmer_len", C.c_uint32), ("min_overlap", C.c_uint32),
                ("min_call_coverage", C.c_int32), ("min_identity", C.c_double)]


class MsaResult(C.Structure):
    _fields_ = [("n_rows", C.c_uint32), ("n_retrieved", C.c_uint32), ("cons_len", C.c_uint32), ("rows_by_step_walk", C.c_uint32),
                ("cons_off", C.c_uint64)]


class ReadResult(C.Structure):
    _fields_ = [("merge", C.c_int32), ("n_pieces", C.c_uint32), ("piece_first", C.c_uint64)] + [
        (n, C.c_int64) for n in ("total_reads_len", "corrected_len", "total_seed_num", "total_walk_num", "high_error_num",
                                 "exceed_depth_num", "exceed_leave_num", "fm_num", "dp_num", "seed_dis")] + [("status", C.c_int32), ("pad", C.c_int32)]


class KernelStats(C.Structure):
    _fields_ = [
        ("launches", C.c_uint64),
        ("total_ms", C.c_double),
        ("rank_queries", C.c_uint64),
        ("block_loads", C.c_uint64),
        ("table_loads", C.c_uint64),
    ]


K_RANK, K_FIND, K_GRID, K_SEEDS, K_EXTEND, K_LF, K_DP, K_MSA = range(8)
SEED_DTYPE = np.dtype([("start", "<i4"), ("len", "<i4"), ("max_freq", "<i4"), ("repeat", "<i4"), ("start_k", "<i4"),
                       ("end_k", "<i4"), ("start_freq", "<i4"), ("end_freq", "<i4")])
BWT, RBWT = 0, 1

RANK_DTYPE = np.dtype([("idx", "<i8"), ("base", "u1"), ("strand", "u1"), ("pad", "u1", (6,))])
BIIV_DTYPE = np.dtype([("fwd_lower", "<i8"), ("fwd_upper", "<i8"), ("rvc_lower", "<i8"), ("rvc_upper", "<i8")])
assert RANK_DTYPE.itemsize == C.sizeof(RankQuery) and BIIV_DTYPE.itemsize == C.sizeof(BiInterval)


def declared_symbols(header: Path | None = None) -> list[str]:
    """Every function include/lrsc.h declares (used by the export test)."""
    text = (header or (_REPO / "include" / "lrsc.h")).read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(lrsc_[a-z0-9_]+)\s*\(", text)))


def _ptr(a: np.ndarray, t=C.c_void_p):
    return a.ctypes.data_as(t)


class Lrsc:
    """Thin object wrapper over the C ABI."""

    def __init__(self, path: os.PathLike | None = None):
        p = Path(path) if path else lib_path()
        if not p.exists():
            raise ImportError(
                f"{p} is missing: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()'). "
                "There is no CPU fallback for the product path."
            )
        self.path = p
        self.lib = C.CDLL(str(p))
        L = self.lib
        L.lrsc_strerror.restype = C.c_char_p
        L.lrsc_last_error.restype = C.c_char_p
        for name in declared_symbols():
            getattr(L, name)  # raises AttributeError if the library does not export it
        L.lrsc_index_open.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_void_p)]
        L.lrsc_index_from_units.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64,
                                            C.POINTER(C.c_void_p)]
        L.lrsc_index_info_get.argtypes = [C.c_void_p, C.POINTER(IndexInfo)]
        L.lrsc_index_upload.argtypes = [C.c_void_p, C.c_int]
        L.lrsc_index_close.argtypes = [C.c_void_p]
        L.lrsc_index_close.restype = None
        L.lrsc_build_bwt.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.POINTER(C.c_void_p),
                                     C.POINTER(C.c_uint64)]
        L.lrsc_buffer_free.argtypes = [C.c_void_p]
        L.lrsc_buffer_free.restype = None
        L.lrsc_write_bwt_file.argtypes = [C.c_char_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64]
        L.lrsc_params_default.argtypes = [C.c_int, C.c_int, C.POINTER(Params)]
        L.lrsc_ctx_create.argtypes = [C.c_void_p, C.POINTER(Params), C.c_int, C.POINTER(C.c_void_p)]
        L.lrsc_ctx_destroy.argtypes = [C.c_void_p]
        L.lrsc_ctx_destroy.restype = None
        L.lrsc_ctx_sync.argtypes = [C.c_void_p]
        L.lrsc_rank.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
        L.lrsc_bwt_chars.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint64, C.c_void_p]
        L.lrsc_find_kmers.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint64, C.c_void_p]
        L.lrsc_kmer_grid.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32,
                                     C.c_void_p, C.c_void_p, C.c_void_p]
        L.lrsc_dp_align.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64, C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_int,
                                    C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
        L.lrsc_dp_consensus.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint64,
                                        C.POINTER(C.c_uint64)]
        L.lrsc_batch_correct.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64,
                                         C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.lrsc_batch_create.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p)]
        L.lrsc_batch_destroy.argtypes = [C.c_void_p]
        L.lrsc_batch_destroy.restype = None
        L.lrsc_batch_kmer_grid.argtypes = [C.c_void_p, C.c_void_p]
        L.lrsc_kmer_thresholds.argtypes = [C.c_int, C.c_void_p]
        L.lrsc_batch_find_seeds.argtypes = [C.c_void_p, C.c_void_p]
        L.lrsc_kmer_thresholds_range.argtypes = [C.c_int, C.c_int, C.c_void_p]
        L.lrsc_batch_set_debug.argtypes = [C.c_void_p, C.c_int]
        L.lrsc_batch_outcast_seeds.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
        L.lrsc_batch_repeat_ratio.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.lrsc_batch_walk_log.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]
        L.lrsc_batch_seeds.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64),
                                       C.c_void_p]
        L.lrsc_find_seeds.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint64,
                                      C.POINTER(C.c_uint64), C.c_void_p]
        L.lrsc_extend_walks.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p,
                                        C.c_uint64, C.POINTER(C.c_uint64)]
        L.lrsc_correct_reads.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint64,
                                         C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.lrsc_ctx_get_params.argtypes = [C.c_void_p, C.POINTER(Params)]
        L.lrsc_lf_walk.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64,
                                   C.c_void_p]
        L.lrsc_ctx_stats.argtypes = [C.c_void_p, C.c_int, C.POINTER(KernelStats)]
        L.lrsc_ctx_stats_reset.argtypes = [C.c_void_p]
        # the synthetic workload generator + test hook live in their own library (longreadselfcorrect_amd/testkit), not in the ABI
        self.kit = K = C.CDLL(str(lib_path().with_name("liblrsc_testkit.so")))
        K.lrsc_debug_sort_order.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
        K.lrsc_synth_genome.argtypes = [C.c_uint64, C.c_uint64, C.c_void_p]
        K.lrsc_synth_reads.argtypes = [C.c_uint64, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32,
                                       C.c_double, C.c_double, C.c_double, C.c_void_p, C.c_uint64, C.c_void_p]

    # ---- helpers -------------------------------------------------------------------------
    def check(self, status: int, what: str):
        if status != 0:
            raise LrscError(status, what, f"{self.lib.lrsc_strerror(status).decode()}; {self.lib.lrsc_last_error().decode()}")

    def params_default(self, genome: int = 10, coverage: int = 90) -> Params:
        p = Params()
        self.check(self.lib.lrsc_params_default(genome, coverage, C.byref(p)), "lrsc_params_default")
        return p

    def debug_sort_order(self, keys: np.ndarray) -> np.ndarray:
        keys = np.ascontiguousarray(keys, dtype=np.uint64)
        perm = np.zeros(keys.size, dtype=np.uint32)
        self.check(self.kit.lrsc_debug_sort_order(_ptr(keys), keys.size, _ptr(perm)), "lrsc_debug_sort_order")
        return perm

    def kmer_thresholds(self, coverage: int) -> np.ndarray:
        out = np.zeros((3, 52), dtype=np.float32)
        self.check(self.lib.lrsc_kmer_thresholds(coverage, _ptr(out)), "lrsc_kmer_thresholds")
        return out

    def kmer_thresholds_range(self, coverage: int, end: int) -> np.ndarray:
        out = np.zeros((3, end + 2), dtype=np.float32)
        self.check(self.lib.lrsc_kmer_thresholds_range(coverage, end, _ptr(out)), "lrsc_kmer_thresholds_range")
        return out

    # ---- synthetic data -------------------------------------------------------------------
    def synth_genome(self, seed: int, length: int) -> np.ndarray:
        out = np.empty(length, dtype=np.uint8)
        self.check(self.kit.lrsc_synth_genome(seed, length, _ptr(out)), "lrsc_synth_genome")
        return out

    def synth_reads(self, seed: int, genome: np.ndarray, n_reads: int, tmpl_len: int, first_read: int = 0,
                    p_del: float = 0.045, p_sub: float = 0.015, p_ins: float = 0.09):
        """Returns (bases uint8[total], offsets uint64[n_reads+1])."""
        cap = int(n_reads * tmpl_len * 1.25) + 4096
        while True:
            bases = np.empty(cap, dtype=np.uint8)
            off = np.empty(n_reads + 1, dtype=np.uint64)
            st = self.kit.lrsc_synth_reads(seed, _ptr(genome), genome.size, first_read, n_reads, tmpl_len,
                                           p_del, p_sub, p_ins, _ptr(bases), cap, _ptr(off))
            if st == -6:  # LRSC_ERR_CAPACITY
                cap *= 2
                continue
            self.check(st, "lrsc_synth_reads")
            return bases[: int(off[-1])].copy(), off

    # ---- index construction -------------------------------------------------------------------
    def build_bwt(self, bases: np.ndarray, off: np.ndarray, reverse_reads: bool, device: int = 0) -> np.ndarray:
        """GPU suffix sort -> RL units (uint8) of the .bwt (or .rbwt) payload."""
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        off = np.ascontiguousarray(off, dtype=np.uint64)
        p = C.c_void_p()
        n = C.c_uint64()
        self.check(self.lib.lrsc_build_bwt(_ptr(bases), _ptr(off), off.size - 1, int(reverse_reads), device,
                                           C.byref(p), C.byref(n)), "lrsc_build_bwt")
        try:
            return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(n.value,)).copy()
        finally:
            self.lib.lrsc_buffer_free(p)

    def write_bwt_file(self, path, units: np.ndarray, num_strings: int, num_symbols: int):
        units = np.ascontiguousarray(units, dtype=np.uint8)
        self.check(self.lib.lrsc_write_bwt_file(str(path).encode(), _ptr(units), units.size, num_strings, num_symbols),
                   "lrsc_write_bwt_file")

    # ---- index / ctx ------------------------------------------------------------------------
    def index_open(self, bwt_path: str, rbwt_path: str) -> "Index":
        h = C.c_void_p()
        self.check(self.lib.lrsc_index_open(str(bwt_path).encode(), str(rbwt_path).encode(), C.byref(h)), "lrsc_index_open")
        return Index(self, h)

    def index_from_units(self, bwt_units: np.ndarray, rbwt_units: np.ndarray, num_strings: int, num_symbols: int) -> "Index":
        h = C.c_void_p()
        a = np.ascontiguousarray(bwt_units, dtype=np.uint8)
        b = np.ascontiguousarray(rbwt_units, dtype=np.uint8)
        self.check(self.lib.lrsc_index_from_units(_ptr(a), a.size, _ptr(b), b.size, num_strings, num_symbols, C.byref(h)),
                   "lrsc_index_from_units")
        return Index(self, h)


class Index:
    def __init__(self, api: Lrsc, handle):
        self.api, self.h = api, handle

    def info(self) -> IndexInfo:
        out = IndexInfo()
        self.api.check(self.api.lib.lrsc_index_info_get(self.h, C.byref(out)), "lrsc_index_info_get")
        return out

    def upload(self, device: int = 0):
        self.api.check(self.api.lib.lrsc_index_upload(self.h, device), "lrsc_index_upload")

    def ctx(self, params: Params | None = None, device: int = 0) -> "Ctx":
        h = C.c_void_p()
        pp = C.byref(params) if params is not None else None
        self.api.check(self.api.lib.lrsc_ctx_create(self.h, pp, device, C.byref(h)), "lrsc_ctx_create")
        return Ctx(self.api, self, h)

    def close(self):
        if self.h:
            self.api.lib.lrsc_index_close(self.h)
            self.h = None


class Ctx:
    def __init__(self, api: Lrsc, index: Index, handle):
        self.api, self.index, self.h = api, index, handle

    def close(self):
        if self.h:
            self.api.lib.lrsc_ctx_destroy(self.h)
            self.h = None

    def rank(self, bases: np.ndarray, idx: np.ndarray, strand: np.ndarray | int) -> np.ndarray:
        n = len(idx)
        q = np.zeros(n, dtype=RANK_DTYPE)
        q["idx"] = idx
        q["base"] = bases
        q["strand"] = strand
        out = np.empty(n, dtype=np.uint64)
        self.api.check(self.api.lib.lrsc_rank(self.h, _ptr(q), n, _ptr(out)), "lrsc_rank")
        return out

    def bwt_chars(self, strand: int, idx: np.ndarray) -> np.ndarray:
        idx = np.ascontiguousarray(idx, dtype=np.uint64)
        out = np.empty(idx.size, dtype=np.uint8)
        self.api.check(self.api.lib.lrsc_bwt_chars(self.h, strand, _ptr(idx), idx.size, _ptr(out)), "lrsc_bwt_chars")
        return out

    def find_kmers(self, kmers: np.ndarray, k: int) -> np.ndarray:
        kmers = np.ascontiguousarray(kmers, dtype=np.uint8)
        n = kmers.size // k
        out = np.empty(n, dtype=BIIV_DTYPE)
        self.api.check(self.api.lib.lrsc_find_kmers(self.h, _ptr(kmers), k, n, _ptr(out)), "lrsc_find_kmers")
        return out

    def kmer_grid(self, bases: np.ndarray, off: np.ndarray, ks, want_iv=True, want_size=True, want_count=True):
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        off = np.ascontiguousarray(off, dtype=np.uint64)
        ks = np.ascontiguousarray(ks, dtype=np.uint8)
        total = int(off[-1]) if off.size else 0
        recs = total * ks.size
        iv = np.empty(recs, dtype=BIIV_DTYPE) if want_iv else None
        size = np.empty(recs, dtype=np.uint8) if want_size else None
        cnt = np.empty((recs, 4), dtype=np.uint8) if want_count else None
        self.api.check(
            self.api.lib.lrsc_kmer_grid(self.h, _ptr(bases), _ptr(off), off.size - 1, _ptr(ks), ks.size,
                                        _ptr(iv) if iv is not None else None,
                                        _ptr(size) if size is not None else None,
                                        _ptr(cnt) if cnt is not None else None),
            "lrsc_kmer_grid")
        shape = (total, ks.size)
        return (iv.reshape(shape) if iv is not None else None,
                size.reshape(shape) if size is not None else None,
                cnt.reshape(shape + (4,)) if cnt is not None else None)

    def lf_walk(self, rows, strand, max_steps) -> list[str]:
        rows = np.ascontiguousarray(rows, dtype=np.uint64)
        strand = np.ascontiguousarray(strand, dtype=np.uint8)
        max_steps = np.ascontiguousarray(max_steps, dtype=np.uint32)
        off = np.zeros(rows.size, dtype=np.uint64)
        off[1:] = np.cumsum(max_steps[:-1].astype(np.uint64))
        cap = int(max_steps.astype(np.uint64).sum()) + 1
        out = np.zeros(cap, dtype=np.uint8)
        lens = np.zeros(rows.size, dtype=np.uint32)
        self.api.check(self.api.lib.lrsc_lf_walk(self.h, _ptr(rows), _ptr(strand), _ptr(max_steps), _ptr(off), rows.size, _ptr(out), cap,
                                                 _ptr(lens)), "lrsc_lf_walk")
        buf = out.tobytes()
        return [buf[int(off[i]): int(off[i]) + int(lens[i])].decode() for i in range(rows.size)]

    def dp_align(self, pairs, band_width=200, scores=(1, -1, -8)):
        """pairs: list of (s1, s2, start1, start2) -> list of dicts like the oracle's extend_match (compact cigar)."""
        n = len(pairs)
        jobs = (DpJob * n)()
        parts, off = [], 0
        for i, (s1, s2, a, b) in enumerate(pairs):
            j = jobs[i]
            j.s1_off, j.s1_len = off, len(s1); off += len(s1)
            j.s2_off, j.s2_len = off, len(s2); off += len(s2)
            j.start1, j.start2 = a, b
            parts += [s1, s2]
        seq = "".join(parts).encode()
        res = (DpResult * n)()
        cap = len(seq) + n + 64
        arena = C.create_string_buffer(cap)
        used = C.c_uint64()
        self.api.check(self.api.lib.lrsc_dp_align(self.h, seq, len(seq), jobs, n, band_width, scores[0], scores[1], scores[2], res,
                                                  arena, cap, C.byref(used)), "lrsc_dp_align")
        out = []
        for r in res:
            ops = arena.raw[r.cigar_off: r.cigar_off + r.cigar_len].decode()
            cig, k = [], 0
            while k < len(ops):
                e = k
                while e < len(ops) and ops[e] == ops[k]:
                    e += 1
                cig.append(f"{e - k}{ops[k]}")
                k = e
            out.append(dict(m0s=r.match0_start, m0e=r.match0_end, m1s=r.match1_start, m1e=r.match1_end, score=r.score,
                            edit=r.edit_distance, cols=r.total_columns, cigar="".join(cig)))
        return out

    def dp_consensus(self, queries):
        """queries: list of (query, kmer_len, min_overlap, min_identity, min_call_coverage) -> list of (rows, consensus, retrieved)."""
        n = len(queries)
        qs = (MsaQuery * n)()
        off = 0
        for i, (q, k, mo, mi, mc) in enumerate(queries):
            qs[i].seq_off, qs[i].len, qs[i].kmer_len, qs[i].min_overlap, qs[i].min_identity, qs[i].min_call_coverage = off, len(q), k, mo, mi, mc
            off += len(q)
        seq = "".join(q[0] for q in queries).encode()
        res = (MsaResult * n)()
        cap = 2 * len(seq) + 256 * n + 64
        arena = C.create_string_buffer(cap)
        used = C.c_uint64()
        self.api.check(self.api.lib.lrsc_dp_consensus(self.h, seq, len(seq), qs, n, res, arena, cap, C.byref(used)), "lrsc_dp_consensus")
        self.msa_rows_by_step_walk = sum(r.rows_by_step_walk for r in res)      # diagnostic of the last call
        return [(r.n_rows, arena.raw[r.cons_off: r.cons_off + r.cons_len].decode(), r.n_retrieved) for r in res]

    def extend_walks(self, walks):
        """walks: list of (src, path, trg, dis, init_kmer, max_overlap, min_sa).  -> list of (code, mergedSeq, steps)."""
        n = len(walks)
        descs = (WalkDesc * n)()
        parts, off = [], 0
        for i, (src, path, trg, dis, initk, maxov, minsa) in enumerate(walks):
            d = descs[i]
            d.seq_off, d.src_len, d.path_len, d.trg_len = off, len(src), len(path), len(trg)
            d.dis, d.init_kmer, d.max_overlap, d.min_sa_threshold = dis, initk, maxov, minsa
            parts += [src, path, trg]
            off += len(src) + len(path) + len(trg)
        seq = "".join(parts).encode()
        res = (WalkResult * n)()
        cap = max(1 << 16, 2 * len(seq) + 4096)
        used = C.c_uint64()
        while True:
            arena = C.create_string_buffer(cap)
            st = self.api.lib.lrsc_extend_walks(self.h, seq, len(seq), descs, n, res, arena, cap, C.byref(used))
            if st == -6 and used.value > cap:
                cap = int(used.value)
                continue
            self.api.check(st, "lrsc_extend_walks")
            break
        raw = arena.raw
        return [(r.code, raw[r.out_off: r.out_off + r.out_len].decode() if r.code > 0 else "", r.steps) for r in res]

    def correct_reads(self, bases: np.ndarray, off: np.ndarray):
        """The whole per-read path.  -> (results: list[ReadResult], pieces: list[list[str]])."""
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        off = np.ascontiguousarray(off, dtype=np.uint64)
        n = off.size - 1
        res = (ReadResult * n)()
        pcap, ocap = 2 * n + 16, int(off[-1]) * 2 + 4096
        npieces, used = C.c_uint64(), C.c_uint64()
        while True:
            poff = np.zeros(pcap + 1, dtype=np.uint64)
            out = np.zeros(ocap, dtype=np.uint8)
            st = self.api.lib.lrsc_correct_reads(self.h, _ptr(bases), _ptr(off), n, res, _ptr(poff), pcap + 1, _ptr(out), ocap,
                                                 C.byref(npieces), C.byref(used))
            if st == -6 and (int(npieces.value) + 1 > pcap or int(used.value) > ocap):
                pcap, ocap = max(pcap, int(npieces.value) + 1), max(ocap, int(used.value))
                continue
            self.api.check(st, "lrsc_correct_reads")
            break
        buf = out.tobytes()
        pieces = []
        for r in res:
            pieces.append([buf[int(poff[p]): int(poff[p + 1])].decode() for p in range(r.piece_first, r.piece_first + r.n_pieces)])
        return list(res), pieces

    def batch(self, bases: np.ndarray, off: np.ndarray) -> "Batch":
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        off = np.ascontiguousarray(off, dtype=np.uint64)
        h = C.c_void_p()
        self.api.check(self.api.lib.lrsc_batch_create(self.h, _ptr(bases), _ptr(off), off.size - 1, C.byref(h)),
                       "lrsc_batch_create")
        return Batch(self, h, int(off[-1]), off.size - 1)

    def stats(self, kernel: int) -> KernelStats:
        s = KernelStats()
        self.api.check(self.api.lib.lrsc_ctx_stats(self.h, kernel, C.byref(s)), "lrsc_ctx_stats")
        return s

    def stats_reset(self):
        self.api.check(self.api.lib.lrsc_ctx_stats_reset(self.h), "lrsc_ctx_stats_reset")

    def sync(self):
        self.api.check(self.api.lib.lrsc_ctx_sync(self.h), "lrsc_ctx_sync")


class Batch:
    def __init__(self, ctx: Ctx, handle, total_bases: int, n_reads: int):
        self.ctx, self.h, self.total_bases, self.n_reads = ctx, handle, total_bases, n_reads

    def kmer_grid(self):
        self.ctx.api.check(self.ctx.api.lib.lrsc_batch_kmer_grid(self.ctx.h, self.h), "lrsc_batch_kmer_grid")

    def find_seeds(self):
        self.ctx.api.check(self.ctx.api.lib.lrsc_batch_find_seeds(self.ctx.h, self.h), "lrsc_batch_find_seeds")

    def seeds(self, want_attribute: bool = True):
        """-> (seed_count uint32[n_reads], seeds SEED_DTYPE[n], attribute int8[total] | None)."""
        api = self.ctx.api
        count = np.zeros(self.n_reads, dtype=np.uint32)
        n = C.c_uint64()
        attr = np.zeros(self.total_bases, dtype=np.int8) if want_attribute else None
        cap = max(1024, self.total_bases // 8)
        while True:
            seeds = np.zeros(cap, dtype=SEED_DTYPE)
            st = api.lib.lrsc_batch_seeds(self.ctx.h, self.h, _ptr(count), _ptr(seeds), cap, C.byref(n),
                                          _ptr(attr) if attr is not None else None)
            if st == -6:
                cap = int(n.value)
                continue
            api.check(st, "lrsc_batch_seeds")
            return count, seeds[: n.value].copy(), attr

    # --debugseed / --onlyseed diagnostics (include/lrsc.h: LRSC_DEBUG_*)
    DEBUG_OUTCASTS, DEBUG_WALKS, DEBUG_RATIO = 1, 2, 4

    def set_debug(self, flags: int):
        self.ctx.api.check(self.ctx.api.lib.lrsc_batch_set_debug(self.h, flags), "lrsc_batch_set_debug")

    def outcast_seeds(self):
        """-> (outcast_count uint32[n_reads], seeds SEED_DTYPE[n]): what removeHitchhikingSeeds dropped."""
        api = self.ctx.api
        count = np.zeros(self.n_reads, dtype=np.uint32)
        n = C.c_uint64()
        api.check(api.lib.lrsc_batch_outcast_seeds(self.ctx.h, self.h, _ptr(count), None, 2 ** 62, C.byref(n)), "lrsc_batch_outcast_seeds")
        seeds = np.zeros(max(1, n.value), dtype=SEED_DTYPE)
        api.check(api.lib.lrsc_batch_outcast_seeds(self.ctx.h, self.h, _ptr(count), _ptr(seeds), seeds.size, C.byref(n)),
                  "lrsc_batch_outcast_seeds")
        return count, seeds[: n.value].copy()

    def repeat_ratio(self) -> np.ndarray:
        out = np.zeros(self.total_bases, dtype=np.float32)
        self.ctx.api.check(self.ctx.api.lib.lrsc_batch_repeat_ratio(self.ctx.h, self.h, _ptr(out)), "lrsc_batch_repeat_ratio")
        return out

    def walk_log(self, n_seeds: int) -> np.ndarray:
        out = np.zeros(max(1, n_seeds), dtype=np.uint8)
        self.ctx.api.check(self.ctx.api.lib.lrsc_batch_walk_log(self.ctx.h, self.h, _ptr(out), out.size), "lrsc_batch_walk_log")
        return out[:n_seeds]

    def correct(self):
        """lrsc_batch_correct on the resident batch -> (results ReadResult[n], piece offsets uint64[], corrected bytes uint8[])."""
        api = self.ctx.api
        n = self.n_reads
        res = (ReadResult * n)()
        if not hasattr(self, "_poff"):
            self._poff = np.zeros(2 * n + 17, dtype=np.uint64)
            self._out = np.zeros(self.total_bases * 2 + 4096, dtype=np.uint8)
        npieces, used = C.c_uint64(), C.c_uint64()
        while True:
            st = api.lib.lrsc_batch_correct(self.ctx.h, self.h, res, _ptr(self._poff), self._poff.size, _ptr(self._out), self._out.size,
                                            C.byref(npieces), C.byref(used))
            if st == -6 and (int(npieces.value) + 1 > self._poff.size or int(used.value) > self._out.size):
                self._poff = np.zeros(max(self._poff.size, int(npieces.value) + 1), dtype=np.uint64)
                self._out = np.zeros(max(self._out.size, int(used.value)), dtype=np.uint8)
                continue
            api.check(st, "lrsc_batch_correct")
            return res, self._poff[: int(npieces.value) + 1], self._out[: int(used.value)]

    def close(self):
        if self.h:
            self.ctx.api.lib.lrsc_batch_destroy(self.h)
            self.h = None
