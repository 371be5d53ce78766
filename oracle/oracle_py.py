"""ctypes loaders for the CPU checker libraries -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

  Oracle  -> oracle/_build/liblrsc_oracle.so   (our CPU restatement of the reference algorithm)
  Ref     -> oracle/_ref/liblrsc_ref.so        (the reference's own sources, compiled by oracle/Makefile;
                                                present only where /root/reference was available at build time)
"""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

BIIV_DTYPE = np.dtype([("fwd_lower", "<i8"), ("fwd_upper", "<i8"), ("rvc_lower", "<i8"), ("rvc_upper", "<i8")])
SEED_FIELDS = ("start", "len", "max_freq", "repeat", "start_k", "end_k", "start_freq", "end_freq")

HERE = Path(__file__).resolve().parent
ORACLE_SO = HERE / "_build" / "liblrsc_oracle.so"
REF_SO = HERE / "_ref" / "liblrsc_ref.so"


def build_oracle():
    subprocess.run(["make", "-C", str(HERE), "oracle"], check=True, capture_output=True)


def build_ref(reference: str = "/root/reference") -> bool:
    if not Path(reference).is_dir():
        return REF_SO.exists()
    subprocess.run(["make", "-C", str(HERE), "ref", "-j8", f"REF={reference}"], check=True, capture_output=True)
    return REF_SO.exists()


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def pack_reads(reads) -> tuple[np.ndarray, np.ndarray]:
    """list[str|bytes] -> (uint8 bases, uint64 offsets)."""
    bs = [r.encode() if isinstance(r, str) else bytes(r) for r in reads]
    off = np.zeros(len(bs) + 1, dtype=np.uint64)
    if bs:
        off[1:] = np.cumsum([len(b) for b in bs])
    bases = np.frombuffer(b"".join(bs), dtype=np.uint8).copy() if bs else np.zeros(0, dtype=np.uint8)
    return bases, off


def unpack_reads(bases: np.ndarray, off: np.ndarray) -> list[str]:
    buf = bases.tobytes()
    return [buf[int(off[i]): int(off[i + 1])].decode() for i in range(len(off) - 1)]


class Oracle:
    def __init__(self):
        if not ORACLE_SO.exists():
            build_oracle()
        self.lib = L = C.CDLL(str(ORACLE_SO))
        L.orc_last_error.restype = C.c_char_p
        L.orc_bwt_load.restype = C.c_void_p
        L.orc_bwt_load.argtypes = [C.c_char_p]
        L.orc_bwt_free.argtypes = [C.c_void_p]
        L.orc_bwt_from_units.restype = C.c_void_p
        L.orc_bwt_from_units.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64]
        for f in ("orc_bwt_num_strings", "orc_bwt_num_symbols", "orc_bwt_num_runs", "orc_bwt_occ_calls"):
            getattr(L, f).restype = C.c_uint64
            getattr(L, f).argtypes = [C.c_void_p]
        L.orc_bwt_pc.restype = C.c_uint64
        L.orc_bwt_pc.argtypes = [C.c_void_p, C.c_char]
        L.orc_bwt_occ_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
        L.orc_bwt_char_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
        L.orc_bwt_decode.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_find_intervals.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint64, C.c_void_p]
        L.orc_build_bwt_file.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_char_p]
        L.orc_threshold_table.argtypes = [C.c_int, C.c_void_p]
        L.orc_threshold_text.argtypes = [C.c_int, C.c_char_p, C.c_int]
        L.orc_kmer_grid.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32,
                                    C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_find_seeds.restype = C.c_int64
        L.orc_find_seeds.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32,
                                     C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]

    # ---- interval tree / DP / FM-extend / whole path -----------------------------------------------
    def _decl_late(self):
        L = self.lib
        if getattr(self, "_late", False):
            return
        self._late = True
        L.orc_itree_build.restype = C.c_void_p
        L.orc_itree_build.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]
        L.orc_itree_free.argtypes = [C.c_void_p]
        L.orc_itree_query.restype = C.c_uint64
        L.orc_itree_query.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_uint64]
        L.orc_extend_match.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                       C.c_void_p, C.c_char_p, C.c_int]
        L.orc_dp_consensus.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int, C.c_char_p,
                                       C.c_int, C.c_void_p]
        L.orc_dp_consensus.restype = C.c_int
        L.orc_extend_walk.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_int,
                                      C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_int, C.c_void_p]
        L.orc_correct_reads.restype = C.c_void_p
        L.orc_correct_reads.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_char_p]
        L.orc_run_free.argtypes = [C.c_void_p]
        L.orc_run_text.restype = C.c_uint64
        L.orc_run_text.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint64]
        L.orc_run_counters.restype = C.c_uint64
        L.orc_run_counters.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
        L.orc_run_walks.restype = C.c_uint64
        L.orc_run_walks.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
        L.orc_run_walk_stats.argtypes = [C.c_void_p, C.c_void_p]

    def itree_query_all(self, start, stop, value, queries):
        """Build the tree over (start, stop, value) and return, per query (qs, qe), the values findOverlapping yields in order."""
        self._decl_late()
        return _itree_query_all(self.lib, "orc", start, stop, value, queries)

    def extend_match(self, s1: str, s2: str, start1: int, start2: int, bandwidth: int = 200, scores=(1, -1, -8)):
        self._decl_late()
        return _extend_match(self.lib.orc_extend_match, s1, s2, start1, start2, bandwidth, scores)

    def dp_consensus(self, bwt, rbwt, query: str, k: int, min_overlap: int, min_identity: float, coverage: int, min_call_coverage: int):
        """buildMultipleAlignment + calculateBaseConsensus -> (rows, consensus, retrieved strings)."""
        self._decl_late()
        out = C.create_string_buffer(4 * len(query) + 1024)
        n3 = np.zeros(3, dtype=np.int32)
        rows = self.lib.orc_dp_consensus(bwt.h, rbwt.h, query.encode(), k, min_overlap, min_identity, coverage, min_call_coverage, out,
                                         len(out), _p(n3))
        assert rows >= 0
        return rows, out.value.decode(), int(n3[0] + n3[1])

    def extend_walk(self, bwt, rbwt, params, src: str, path: str, trg: str, dis: int, initk: int, max_overlap: int,
                    min_sa: int):
        """One LongReadSelfCorrectByOverlap walk -> (code, mergedSeq, (steps, leaf_expansions, refine_calls))."""
        self._decl_late()
        cap = len(src) + 3 * len(path) + len(trg) + 4096
        out = C.create_string_buffer(cap)
        st = np.zeros(3, dtype=np.uint64)
        code = self.lib.orc_extend_walk(bwt.h, rbwt.h, C.byref(params), src.encode(), path.encode(), trg.encode(), dis,
                                        initk, max_overlap, min_sa, out, cap, _p(st))
        return code, out.value.decode(), tuple(int(x) for x in st)

    def correct_reads(self, bwt, rbwt, params, bases, off, id_prefix: str = "r") -> "OracleRun":
        self._decl_late()
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        off = np.ascontiguousarray(off, dtype=np.uint64)
        h = self.lib.orc_correct_reads(bwt.h, rbwt.h, C.byref(params), _p(bases), _p(off), off.size - 1, id_prefix.encode())
        return OracleRun(self, h)

    def bwt_load(self, path) -> "OracleBwt":
        h = self.lib.orc_bwt_load(str(path).encode())
        if not h:
            raise RuntimeError(self.lib.orc_last_error().decode())
        return OracleBwt(self, h)

    def bwt_from_units(self, units: np.ndarray, num_strings: int, num_symbols: int) -> "OracleBwt":
        units = np.ascontiguousarray(units, dtype=np.uint8)
        return OracleBwt(self, self.lib.orc_bwt_from_units(_p(units), units.size, num_strings, num_symbols))

    def build_bwt_file(self, bases: np.ndarray, off: np.ndarray, reverse_reads: bool, out_path):
        st = self.lib.orc_build_bwt_file(_p(bases), _p(off), len(off) - 1, int(reverse_reads), str(out_path).encode())
        if st != 0:
            raise RuntimeError(self.lib.orc_last_error().decode())

    def build_index(self, bases, off, prefix):
        """Writes <prefix>.bwt and <prefix>.rbwt."""
        self.build_bwt_file(bases, off, False, f"{prefix}.bwt")
        self.build_bwt_file(bases, off, True, f"{prefix}.rbwt")

    def threshold_table(self, cov: int) -> np.ndarray:
        out = np.zeros((3, 52), dtype=np.float32)
        self.lib.orc_threshold_table(cov, _p(out))
        return out

    def threshold_text(self, cov: int) -> str:
        buf = C.create_string_buffer(8192)
        n = self.lib.orc_threshold_text(cov, buf, 8192)
        assert n >= 0
        return buf.value.decode()

    def kmer_grid(self, bwt: "OracleBwt", rbwt: "OracleBwt", bases, off, ks, outputs: bool = True):
        """KmerFeature grid; same record layout as lrsc_kmer_grid -> (iv[total,n_k] BIIV, size, count[...,4])."""
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        off = np.ascontiguousarray(off, dtype=np.uint64)
        ks = np.ascontiguousarray(ks, dtype=np.uint8)
        total = int(off[-1])
        if not outputs:      # timing only (bench.py cpu_baseline)
            self.lib.orc_kmer_grid(bwt.h, rbwt.h, _p(bases), _p(off), off.size - 1, _p(ks), ks.size, None, None, None)
            return None
        iv = np.zeros((total, ks.size), dtype=BIIV_DTYPE)
        size = np.zeros((total, ks.size), dtype=np.uint8)
        cnt = np.zeros((total, ks.size, 4), dtype=np.uint8)
        self.lib.orc_kmer_grid(bwt.h, rbwt.h, _p(bases), _p(off), off.size - 1, _p(ks), ks.size, _p(iv), _p(size), _p(cnt))
        return iv, size, cnt

    def find_seeds_debug(self, bwt: "OracleBwt", rbwt: "OracleBwt", params, bases, off):
        """What --debugseed dumps beside the seeds -> (outcast_count uint32[n_reads], outcasts int32[n,8], ratio float32[total])."""
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        off = np.ascontiguousarray(off, dtype=np.uint64)
        n_reads = off.size - 1
        cap = max(1024, int(off[-1]) // 8)
        count = np.zeros(n_reads, dtype=np.uint32)
        seeds = np.zeros((cap, 8), dtype=np.int32)
        ratio = np.zeros(int(off[-1]), dtype=np.float32)
        self.lib.orc_find_seeds_debug.restype = C.c_int64
        n = self.lib.orc_find_seeds_debug(C.c_void_p(bwt.h), C.c_void_p(rbwt.h), C.byref(params), _p(bases), _p(off), C.c_uint32(n_reads),
                                          _p(count), _p(seeds), C.c_uint64(cap), _p(ratio))
        if n < 0:
            raise RuntimeError("seed capacity too small")
        return count, seeds[:n].copy(), ratio

    def saipb_merge(self, bwt, rbwt, source: str, between: str, target: str, dis: int, max_leaves: int = 32):
        """SAIPBSelfCorrectTree driven like its (commented-out) call site -> (code, merged, stats dict)."""
        cap = len(source) + 3 * max(dis, 0) + len(target) + 4096
        out = C.create_string_buffer(cap)
        st = (C.c_int64 * 6)()
        self.lib.orc_saipb_merge.restype = C.c_int
        rc = self.lib.orc_saipb_merge(C.c_void_p(bwt.h), C.c_void_p(rbwt.h), source.encode(), between.encode(), target.encode(), C.c_int(dis),
                                      C.c_int(max_leaves), out, C.c_uint64(cap), st)
        keys = ("steps", "max_leaves", "results", "hash_entries", "source_freq", "target_freq")
        return rc, out.value.decode(), dict(zip(keys, [int(x) for x in st]))

    def stdaln_global(self, s1: str, s2: str):
        """aln_stdaln(s1, s2, &aln_param_pacbio, GLOBAL, 1) -> ('|' count, score, path_len)."""
        out = (C.c_int * 3)()
        self.lib.orc_stdaln_global(s1.encode(), s2.encode(), out)
        return int(out[0]), int(out[1]), int(out[2])

    def threshold_table_range(self, cov: int, end: int) -> np.ndarray:
        out = np.zeros((3, end + 2), dtype=np.float32)
        self.lib.orc_threshold_table_range(cov, end, _p(out))
        return out

    def find_seeds(self, bwt: "OracleBwt", rbwt: "OracleBwt", params, bases, off):
        """-> (seed_count uint32[n_reads], seeds int32[n,8], attribute int8[total])."""
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        off = np.ascontiguousarray(off, dtype=np.uint64)
        n_reads = off.size - 1
        cap = max(1024, int(off[-1]) // 8)
        count = np.zeros(n_reads, dtype=np.uint32)
        seeds = np.zeros((cap, 8), dtype=np.int32)
        attr = np.zeros(int(off[-1]), dtype=np.int8)
        n = self.lib.orc_find_seeds(bwt.h, rbwt.h, C.byref(params), _p(bases), _p(off), n_reads, _p(count), _p(seeds), cap,
                                    _p(attr))
        if n < 0:
            raise RuntimeError("seed capacity too small")
        return count, seeds[:n].copy(), attr


def _itree_query_all(lib, prefix, start, stop, value, queries):
    start = np.ascontiguousarray(start, dtype=np.uint64)
    stop = np.ascontiguousarray(stop, dtype=np.uint64)
    value = np.ascontiguousarray(value, dtype=np.uint64)
    build, free, query = (getattr(lib, f"{prefix}_itree_{f}") for f in ("build", "free", "query"))
    h = build(_p(start), _p(stop), _p(value), start.size)
    out = []
    buf = np.zeros(max(16, start.size), dtype=np.uint64)
    for qs, qe in queries:
        n = query(h, int(qs), int(qe), _p(buf), buf.size)
        out.append(buf[:n].copy())
    free(h)
    return out


def _extend_match(fn, s1, s2, start1, start2, bandwidth, scores):
    out7 = np.zeros(7, dtype=np.int32)
    cig = C.create_string_buffer(4 * (len(s1) + len(s2)) + 64)
    fn(s1.encode(), s2.encode(), start1, start2, bandwidth, scores[0], scores[1], scores[2], _p(out7), cig, len(cig))
    keys = ("m0s", "m0e", "m1s", "m1e", "score", "edit", "cols")
    d = dict(zip(keys, (int(x) for x in out7)))
    d["cigar"] = cig.value.decode()
    return d


class OracleRun:
    """Result of the whole per-read path over a batch (correct.fa, discard.fa, stats, counters, walks)."""
    COUNTERS = ("totalReadsLen", "correctedLen", "totalSeedNum", "totalWalkNum", "highErrorNum", "exceedDepthNum",
                "exceedLeaveNum", "FMNum", "DPNum", "seedDis", "merge")

    def __init__(self, o: "Oracle", h):
        self.o, self.h = o, h

    def _text(self, which):
        n = self.o.lib.orc_run_text(self.h, which, None, 0)
        buf = C.create_string_buffer(n + 1)
        self.o.lib.orc_run_text(self.h, which, buf, n)
        return buf.raw[:n].decode()

    @property
    def correct_fa(self):
        return self._text(0)

    @property
    def discard_fa(self):
        return self._text(1)

    @property
    def stats(self):
        return self._text(2)

    @property
    def counters(self) -> np.ndarray:
        n = self.o.lib.orc_run_counters(self.h, None, 0)
        out = np.zeros(n, dtype=np.int64)
        self.o.lib.orc_run_counters(self.h, _p(out), n)
        return out.reshape(-1, 11)

    @property
    def walks(self) -> np.ndarray:
        n = self.o.lib.orc_run_walks(self.h, None, 0)
        out = np.zeros(n, dtype=np.int32)
        self.o.lib.orc_run_walks(self.h, _p(out), n)
        return out.reshape(-1, 5)          # read, srcStart, trgStart, code, via(0 FM, 1 DP, 2 raw/split)

    @property
    def walk_work(self) -> np.ndarray:
        f = self.o.lib.orc_run_walk_work
        f.restype = C.c_uint64
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
        n = f(self.h, None, 0)
        out = np.zeros(n, dtype=np.int32)
        f(self.h, _p(out), n)
        return out.reshape(-1, 3)          # gap, extension steps, leaf expansions (same order as walks)

    @property
    def walk_stats(self):
        out = np.zeros(3, dtype=np.uint64)
        self.o.lib.orc_run_walk_stats(self.h, _p(out))
        return tuple(int(x) for x in out)  # steps, leaf expansions, refine calls

    @property
    def spec_stats(self):
        out = np.zeros(6, dtype=np.uint64)
        self.o.lib.orc_run_spec_stats.argtypes = [C.c_void_p, C.c_void_p]
        self.o.lib.orc_run_spec_stats(self.h, _p(out))
        return dict(zip(("walks", "hit", "miss_after_fm", "miss_after_dp", "miss_after_raw", "miss_k"), (int(x) for x in out)))

    def close(self):
        if self.h:
            self.o.lib.orc_run_free(self.h)
            self.h = None


class OracleBwt:
    def __init__(self, o: Oracle, h):
        self.o, self.h = o, h

    @property
    def num_strings(self):
        return self.o.lib.orc_bwt_num_strings(self.h)

    @property
    def num_symbols(self):
        return self.o.lib.orc_bwt_num_symbols(self.h)

    @property
    def num_runs(self):
        return self.o.lib.orc_bwt_num_runs(self.h)

    @property
    def occ_calls(self):
        return self.o.lib.orc_bwt_occ_calls(self.h)

    def pc(self, b: str) -> int:
        return self.o.lib.orc_bwt_pc(self.h, b.encode())

    def occ(self, bases: np.ndarray, idx: np.ndarray) -> np.ndarray:
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        idx = np.ascontiguousarray(idx, dtype=np.int64)
        out = np.empty(idx.size, dtype=np.uint64)
        self.o.lib.orc_bwt_occ_batch(self.h, _p(bases), _p(idx), idx.size, _p(out))
        return out

    def chars(self, idx: np.ndarray) -> np.ndarray:
        idx = np.ascontiguousarray(idx, dtype=np.uint64)
        out = np.empty(idx.size, dtype=np.uint8)
        self.o.lib.orc_bwt_char_batch(self.h, _p(idx), idx.size, _p(out))
        return out

    def decode(self) -> np.ndarray:
        out = np.empty(self.num_symbols, dtype=np.uint8)
        self.o.lib.orc_bwt_decode(self.h, _p(out))
        return out

    def find_intervals(self, kmers: np.ndarray, k: int) -> np.ndarray:
        kmers = np.ascontiguousarray(kmers, dtype=np.uint8)
        n = kmers.size // k
        out = np.empty((n, 2), dtype=np.int64)
        self.o.lib.orc_find_intervals(self.h, _p(kmers), k, n, _p(out))
        return out

    def close(self):
        if self.h:
            self.o.lib.orc_bwt_free(self.h)
            self.h = None


class Ref:
    """The reference's own object code (only where oracle/_ref was built)."""

    @staticmethod
    def available() -> bool:
        return REF_SO.exists()

    def __init__(self):
        if not REF_SO.exists():
            raise FileNotFoundError(REF_SO)
        self.lib = L = C.CDLL(str(REF_SO))
        L.ref_bwt_load.restype = C.c_void_p
        L.ref_bwt_load.argtypes = [C.c_char_p]
        L.ref_bwt_free.argtypes = [C.c_void_p]
        for f in ("ref_bwt_num_strings", "ref_bwt_num_symbols", "ref_bwt_num_runs"):
            getattr(L, f).restype = C.c_uint64
            getattr(L, f).argtypes = [C.c_void_p]
        L.ref_bwt_pc.restype = C.c_uint64
        L.ref_bwt_pc.argtypes = [C.c_void_p, C.c_char]
        L.ref_bwt_occ_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
        L.ref_bwt_char_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
        L.ref_build_bwt.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int]
        L.ref_threshold_table.argtypes = [C.c_int, C.c_void_p]
        L.ref_itree_build.restype = C.c_void_p
        L.ref_itree_build.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]
        L.ref_itree_free.argtypes = [C.c_void_p]
        L.ref_itree_query.restype = C.c_uint64
        L.ref_itree_query.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_uint64]

        L.ref_extend_match.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                       C.c_void_p, C.c_char_p, C.c_int]

    def itree_query_all(self, start, stop, value, queries):
        return _itree_query_all(self.lib, "ref", start, stop, value, queries)

    def extend_match(self, s1: str, s2: str, start1: int, start2: int, bandwidth: int = 200, scores=(1, -1, -8)):
        return _extend_match(self.lib.ref_extend_match, s1, s2, start1, start2, bandwidth, scores)

    def threshold_table(self, cov: int) -> np.ndarray:
        """One coverage per process: the reference object is an initialise-once singleton."""
        out = np.zeros((3, 52), dtype=np.float32)
        self.lib.ref_threshold_table(cov, _p(out))
        return out

    def bwt_load(self, path) -> "RefBwt":
        return RefBwt(self, self.lib.ref_bwt_load(str(path).encode()))

    def build_index(self, fasta_path, prefix, threads: int = 4):
        self.lib.ref_build_bwt(str(fasta_path).encode(), f"{prefix}.bwt".encode(), threads, 0)
        self.lib.ref_build_bwt(str(fasta_path).encode(), f"{prefix}.rbwt".encode(), threads, 1)


class RefBwt:
    def __init__(self, r: Ref, h):
        self.r, self.h = r, h

    @property
    def num_strings(self):
        return self.r.lib.ref_bwt_num_strings(self.h)

    @property
    def num_symbols(self):
        return self.r.lib.ref_bwt_num_symbols(self.h)

    @property
    def num_runs(self):
        return self.r.lib.ref_bwt_num_runs(self.h)

    def pc(self, b: str) -> int:
        return self.r.lib.ref_bwt_pc(self.h, b.encode())

    def occ(self, bases, idx) -> np.ndarray:
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        idx = np.ascontiguousarray(idx, dtype=np.int64)
        out = np.empty(idx.size, dtype=np.uint64)
        self.r.lib.ref_bwt_occ_batch(self.h, _p(bases), _p(idx), idx.size, _p(out))
        return out

    def chars(self, idx) -> np.ndarray:
        idx = np.ascontiguousarray(idx, dtype=np.uint64)
        out = np.empty(idx.size, dtype=np.uint8)
        self.r.lib.ref_bwt_char_batch(self.h, _p(idx), idx.size, _p(out))
        return out

    def close(self):
        if self.h:
            self.r.lib.ref_bwt_free(self.h)
            self.h = None
