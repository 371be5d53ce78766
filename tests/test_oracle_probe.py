"""CPU checks of the oracle's seed-finding layer (threshold table, k-mer grid, seeds)."""
import json
import subprocess
import sys

import numpy as np
import pytest

from .conftest import GOLDEN, REPO


def test_threshold_table_matches_golden(oracle):
    """tests/golden/threshold_tables.json was produced by the reference's own KmerThreshold.cpp
    (oracle/_ref, script tests/golden/make_golden.py)."""
    gold = json.loads((GOLDEN / "threshold_tables.json").read_text())
    for cov, rows in gold.items():
        got = oracle.threshold_table(int(cov))
        want = np.array(rows, dtype=np.float32)
        assert got.tobytes() == want.tobytes(), f"coverage {cov}"


@pytest.mark.parametrize("cov", [30, 90])
def test_threshold_table_matches_reference_object_code(ref, oracle, cov):
    # the reference singleton can be initialised once per process -> one subprocess per coverage
    code = ("import sys; sys.path.insert(0, %r); from oracle import oracle_py as o; import numpy as np;"
            "sys.stdout.buffer.write(o.Ref().threshold_table(%d).tobytes())" % (str(REPO), cov))
    raw = subprocess.run([sys.executable, "-c", code], check=True, capture_output=True).stdout
    assert raw == oracle.threshold_table(cov).tobytes()


def test_threshold_text_shape(oracle):
    txt = oracle.threshold_text(90).splitlines()
    assert txt[0] == "Coverage : 90" and txt[1] == "size\tlowcov\tunique\trepeat"
    assert len(txt) == 2 + 36 and txt[2].startswith("15\t") and txt[-1].startswith("50\t")


def test_kmer_grid_consistent_with_find_interval(oracle, small_ds):
    """Grid slot k at position p == findBiInterval of the k-mer at p whenever no early exit interferes,
    and sizes/counts follow KmerFeature's rules near the read end."""
    ob, orb = oracle.bwt_load(small_ds.prefix + ".bwt"), oracle.bwt_load(small_ds.prefix + ".rbwt")
    bases, off = small_ds.bases, small_ds.off
    sub_off = off[:4].copy()
    sub = bases[: int(sub_off[-1])]
    ks = np.array([5, 9, 15, 17, 19], dtype=np.uint8)
    iv, size, cnt = oracle.kmer_grid(ob, orb, sub, sub_off, ks)
    comp = np.zeros(256, dtype=np.uint8)
    comp[list(b"ACGT")] = list(b"TGCA")
    for r in range(3):
        s, e = int(sub_off[r]), int(sub_off[r + 1])
        L = e - s
        for j, k in enumerate(int(x) for x in ks):
            want_size = np.minimum(k, L - np.arange(L))
            np.testing.assert_array_equal(size[s:e, j], want_size)
            full = np.arange(0, L - k + 1)
            kmers = np.stack([sub[s + p: s + p + k] for p in full])
            fwd = orb.find_intervals(kmers[:, ::-1].copy().reshape(-1), int(k))
            valid = fwd[:, 0] <= fwd[:, 1]
            # where the k-mer exists the chained expand() result equals a from-scratch search
            np.testing.assert_array_equal(iv["fwd_lower"][s:e, j][full][valid], fwd[valid, 0])
            np.testing.assert_array_equal(iv["fwd_upper"][s:e, j][full][valid], fwd[valid, 1])
            # counts == base composition of the k-mer (no early exit when every 5-mer exists)
            comp_cnt = np.stack([(kmers == b).sum(axis=1) for b in b"ACGT"], axis=1)
            np.testing.assert_array_equal(cnt[s:e, j][full], comp_cnt)
    ob.close(); orb.close()


def test_find_seeds_sane(api, oracle, small_ds):
    ob, orb = oracle.bwt_load(small_ds.prefix + ".bwt"), oracle.bwt_load(small_ds.prefix + ".rbwt")
    p = api.params_default(5, 90)
    n = 20
    off = small_ds.off[: n + 1].copy()
    bases = small_ds.bases[: int(off[-1])]
    count, seeds, attr = oracle.find_seeds(ob, orb, p, bases, off)
    assert count.sum() == len(seeds) and count.sum() > n       # several seeds per 2 kb read at 90x
    assert set(np.unique(attr)) <= {1, 2}
    # seeds are disjoint, ordered, inside the read, and at least as long as the static k-mer
    k = 0
    for r in range(n):
        L = int(off[r + 1] - off[r])
        prev_end = -1
        for s in seeds[k: k + count[r]]:
            assert s[0] > prev_end and s[0] + s[1] <= L and s[1] >= 15
            prev_end = s[0] + s[1] - 1
        k += count[r]
    ob.close(); orb.close()
