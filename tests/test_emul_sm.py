"""The correction kernels' per-lane state machine (csrc/walk_sm.h), driven on the CPU by tests/host_emul, against the CPU
oracle: corrected strings and every integer counter, bit for bit -- with no k-mer tables, with 5/9-mer tables, with an
11-mer table on top; default flow (DP answers taken from the oracle's DP restatement) and --nodp; --split; tight budgets
(yield + resume).  This is the same source the GPU kernel compiles; the GPU parity tests run it on the device."""
from __future__ import annotations

import numpy as np
import pytest

from tests.emul import Emul


@pytest.fixture(scope="module")
def emul():
    return Emul()


@pytest.fixture(scope="module")
def ds_units(small_ds):
    u = [np.fromfile(f"{small_ds.prefix}.{ext}", dtype=np.uint8)[30:] for ext in ("bwt", "rbwt")]
    n_sym = int(small_ds.off[-1]) + small_ds.n_reads
    return u, n_sym


def _pieces_fasta(counters, pieces, split):
    out, pi = [], 0
    # reads with merge = 1 own consecutive pieces; without --split exactly one each
    return out, pi


def _run(emul, oracle, api, small_ds, ds_units, tables, nodp, split=0, next_target=1, n_reads=60, wide=False, max_walks=0, max_steps=2000, min_kmer=None):
    (u0, u1), n_sym = ds_units
    h = emul.index(u0, u1, n_sym, wide=wide, tables=tables)
    p = api.params_default(5, 90)
    p.no_dp, p.split, p.next_target = nodp, split, next_target
    if min_kmer:
        p.min_kmer_len = min_kmer
    off = small_ds.off[: n_reads + 1].copy()
    bases = small_ds.bases[: int(off[-1])]
    ob, orb = oracle.bwt_load(small_ds.prefix + ".bwt"), oracle.bwt_load(small_ds.prefix + ".rbwt")
    count, seeds, _ = oracle.find_seeds(ob, orb, p, bases, off)
    want = oracle.correct_reads(ob, orb, p, bases, off)

    def dp(query, k, mo, mi, mc):
        rows, cons, _ = oracle.dp_consensus(ob, orb, query, k, mo, mi, p.pb_coverage, mc)
        return rows, cons

    counters, pieces, stats = emul.correct_reads(h, p, bases, off, count, seeds, dp=None if nodp else dp, max_walks=max_walks, max_steps=max_steps)
    emul.index_free(h)
    np.testing.assert_array_equal(counters, want.counters)
    # correct.fa: pieces in read order
    fa, pi = [], 0
    wl = want.correct_fa.split("\n")
    want_seqs = wl[1::2]
    assert pieces == want_seqs
    want.close(); ob.close(); orb.close()
    return counters, stats


@pytest.mark.parametrize("tables", [(), (5, 9), (5, 9, 11)], ids=["notab", "tab59", "tab5911"])
def test_sm_nodp_matches_oracle(emul, oracle, api, small_ds, ds_units, tables):
    c, stats = _run(emul, oracle, api, small_ds, ds_units, tables, nodp=1)
    assert c[:, 7].sum() > 300 and c[:, 4].sum() > 0


def test_sm_default_flow_matches_oracle(emul, oracle, api, small_ds, ds_units):
    c, stats = _run(emul, oracle, api, small_ds, ds_units, (5, 9, 11), nodp=0)
    assert c[:, 8].sum() > 5 and stats[2] > 1           # DP answers came back through park + resume launches


def test_sm_split_next_target_and_budgets(emul, oracle, api, small_ds, ds_units):
    _run(emul, oracle, api, small_ds, ds_units, (5, 9), nodp=1, split=1, next_target=2, n_reads=40)
    _run(emul, oracle, api, small_ds, ds_units, (5, 9, 11), nodp=0, n_reads=30, max_walks=2, max_steps=150)


def test_sm_wide_layout(emul, oracle, api, small_ds, ds_units):
    _run(emul, oracle, api, small_ds, ds_units, (5, 9), nodp=1, n_reads=30, wide=True)


def test_sm_prep_fast_path(emul, oracle, api, small_ds, ds_units):
    """Tables of exactly the three emitted sizes (5, idmer 9, minOverlap 11 here): PREP answers four offsets per sweep from direct
    table look-ups (on the GPU: 5 / 9 / 13 with the default minOverlap); narrow and wide layouts."""
    c, stats = _run(emul, oracle, api, small_ds, ds_units, (5, 9, 11), nodp=1, n_reads=40, min_kmer=11)
    c2, stats2 = _run(emul, oracle, api, small_ds, ds_units, (5, 9), nodp=1, n_reads=40, min_kmer=11)
    assert stats[0] < stats2[0]                       # fewer sweeps than the one-request-per-sweep path
    _run(emul, oracle, api, small_ds, ds_units, (5, 9, 11), nodp=0, n_reads=20, min_kmer=11, wide=True)


def test_sm_repeat_dataset_reverse_strand_walks(emul, oracle, api, repeat_ds):
    """Repeat-rich reads: isRepeat seeds, repeat-to-unique walks (source and target swapped, everything reverse-complemented,
    PacBioSelfCorrectionProcess.cpp:176-200) -- the stitching of reverse-strand results."""
    u = [np.fromfile(f"{repeat_ds.prefix}.{ext}", dtype=np.uint8)[30:] for ext in ("bwt", "rbwt")]
    n_sym = int(repeat_ds.off[-1]) + repeat_ds.n_reads
    for genome in (5, 10):
        h = emul.index(u[0], u[1], n_sym, tables=(5, 9, 11))
        p = api.params_default(genome, 90)
        p.no_dp = 1
        n = 150
        off = repeat_ds.off[: n + 1].copy()
        bases = repeat_ds.bases[: int(off[-1])]
        ob, orb = oracle.bwt_load(repeat_ds.prefix + ".bwt"), oracle.bwt_load(repeat_ds.prefix + ".rbwt")
        count, seeds, _ = oracle.find_seeds(ob, orb, p, bases, off)
        want = oracle.correct_reads(ob, orb, p, bases, off)
        counters, pieces, _ = emul.correct_reads(h, p, bases, off, count, seeds)
        emul.index_free(h)
        np.testing.assert_array_equal(counters, want.counters)
        assert pieces == want.correct_fa.split("\n")[1::2]
        rev = int(((seeds[:-1, 3] & 1) == 1).sum())
        assert rev > 0                                   # repeat seeds exist in the sample
        want.close(); ob.close(); orb.close()
