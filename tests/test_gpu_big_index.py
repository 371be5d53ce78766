"""A genuine >= 2^32-symbol index in front of the driver (`-m gpu`, about 100 s): BASELINE configs[3]/[4] mechanisms at real size.

420 000 x 10 kb synthetic reads = 4.39 G symbols per strand (1.02 x 2^32): both BWTs by the GPU builder (grouped sorting jobs,
64-bit positions: SuffixTools/BWTCARopebwt.cpp:160-247 is what it replaces), the Block64 upload (RLBWT.h:42-140 semantics through
64-bit counters), LF-walk read-back of 32 sentinel rows on both strands, and seeds + the whole per-read path of a read sample
against the live CPU oracle over the SAME run-length units.  Results also go to gpurun_out/big_index_test.json.
"""
from __future__ import annotations

import json
import time
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
REPO = Path(__file__).resolve().parent.parent
N_READS, READ_LEN, SAMPLE = 420_000, 10_000, 8
NAMES = ("total_reads_len", "corrected_len", "total_seed_num", "total_walk_num", "high_error_num", "exceed_depth_num",
         "exceed_leave_num", "fm_num", "dp_num", "seed_dis", "merge")


def test_index_above_2_32_symbols_build_walk_and_correct(api, oracle):
    rec = {"reads": N_READS, "read_len": READ_LEN}
    genome = api.synth_genome(0xB16, int(N_READS * READ_LEN * 1.045 / 90))
    bases, off = api.synth_reads(0xB17, genome, N_READS, READ_LEN)
    n_sym = int(off[-1]) + N_READS
    assert n_sym >= 2 ** 32, "the point of this test is an index beyond 32-bit positions"
    rec["num_symbols"] = n_sym
    units = []
    for rev in (False, True):
        t = time.time()
        units.append(api.build_bwt(bases, off, rev, 0))
        rec["build_rbwt_s" if rev else "build_bwt_s"] = round(time.time() - t, 2)
    index = api.index_from_units(units[0], units[1], N_READS, n_sym)
    index.upload(0)
    info = index.info()
    assert info.block_symbols == 128 and info.block_bytes == 64, "expected the Block64 layout above 2^31 symbols"
    rec.update(device_bytes=int(info.device_bytes), runs=[int(info.num_runs[0]), int(info.num_runs[1])])
    p = api.params_default(10, 90)
    ctx = index.ctx(p, 0)

    # LF-walk: BWT row r (the r-th sentinel) spells read r backwards; the reversed-read index spells it forwards
    rng = np.random.default_rng(1)
    rows = np.unique(np.concatenate([[0, 1, N_READS - 1], rng.integers(0, N_READS, size=29)])).astype(np.uint64)
    lens = (off[rows.astype(np.int64) + 1] - off[rows.astype(np.int64)]).astype(np.uint32)
    got = ctx.lf_walk(rows, np.zeros(rows.size, dtype=np.uint8), lens + 5)
    for r, s in zip(rows.tolist(), got):
        assert s == bases[int(off[r]): int(off[r + 1])].tobytes().decode()[::-1], f"bwt LF-walk of read {r}"
    got = ctx.lf_walk(rows, np.ones(rows.size, dtype=np.uint8), lens + 5)
    for r, s in zip(rows.tolist(), got):
        assert s == bases[int(off[r]): int(off[r + 1])].tobytes().decode(), f"rbwt LF-walk of read {r}"
    rec["lf_walk_rows"] = int(rows.size)
    rec["lf_walk_rank_steps"] = int(lens.sum()) * 2

    # seeds + whole path of a read sample against the oracle over the same units
    sample = np.unique(np.concatenate([[0, 1], rng.integers(0, N_READS, size=SAMPLE - 2)]))
    sb = np.concatenate([bases[int(off[r]): int(off[r + 1])] for r in sample])
    so = np.zeros(sample.size + 1, dtype=np.uint64)
    so[1:] = np.cumsum([int(off[r + 1] - off[r]) for r in sample])
    ob = oracle.bwt_from_units(units[0], N_READS, n_sym)
    orb = oracle.bwt_from_units(units[1], N_READS, n_sym)
    b = ctx.batch(sb, so)
    b.find_seeds()
    count, seeds, attr = b.seeds()
    wcount, wseeds, wattr = oracle.find_seeds(ob, orb, p, sb, so)
    gs = np.stack([seeds[f] for f in seeds.dtype.names], axis=1).astype(np.int32)
    np.testing.assert_array_equal(count, wcount)
    np.testing.assert_array_equal(gs, wseeds)
    np.testing.assert_array_equal(attr, wattr)
    results, poff, outb = b.correct()
    b.close()
    want = oracle.correct_reads(ob, orb, p, sb, so)
    gc = np.array([[getattr(r, f) for f in NAMES] for r in results], dtype=np.int64)
    buf = outb.tobytes()
    cfa = "".join(f">r{i}\n{buf[int(poff[r.piece_first]): int(poff[r.piece_first + 1])].decode()}\n" for i, r in enumerate(results) if r.merge)
    np.testing.assert_array_equal(gc, want.counters)
    assert cfa == want.correct_fa
    assert all(r.status == 0 for r in results)
    rec.update(sample_reads=int(sample.size), sample_seeds=int(count.sum()), sample_counter_sums=gc.sum(axis=0).tolist())
    want.close(); ob.close(); orb.close(); ctx.close(); index.close()
    out = REPO / "gpurun_out"
    out.mkdir(exist_ok=True)
    (out / "big_index_test.json").write_text(json.dumps(rec, indent=1))
