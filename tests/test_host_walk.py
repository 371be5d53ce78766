"""The product's seed-to-seed walk (csrc/walk_device.h: the source wp_extend_kernel and walk_extend_kernel compile for gfx950),
compiled for the host by tests/host_walk and run on the CPU against the oracle: return code, merged sequence and step count of
every walk -- through Walk::run (general step only) and through the single-leaf fast path with its hand-over to the general step
(the loop of wp_extend_kernel), over the narrow and the wide rank-block layout, with and without k-mer tables.
The GPU parity tests run the same source on the device."""
from __future__ import annotations

import numpy as np
import pytest

from tests.host_walk import HostWalk
from tests.test_gpu_fm import _walk_descs


@pytest.fixture(scope="module")
def hw():
    return HostWalk()


def _units(ds):
    u = [np.fromfile(f"{ds.prefix}.{ext}", dtype=np.uint8)[30:] for ext in ("bwt", "rbwt")]
    return u, int(ds.off[-1]) + ds.n_reads


@pytest.fixture(scope="module")
def ds_units(small_ds):
    return _units(small_ds)


def _skip_descs(params, reads, count, seeds, skip):
    """Walks from seed i to seed i + skip (the seeds in between ignored): gaps of several hundred bases to more than a kilobase,
    most of them beyond what the walk can bridge -- the wide-frontier, failing, many-step walks."""
    descs, k = [], 0
    for r, n in enumerate(count):
        ss = seeds[k: k + n]
        k += n
        read = reads[r]
        for i in range(0, max(int(n) - skip, 0), skip):
            s, t = ss[i], ss[i + skip]
            if s[3] or t[3]:
                continue
            s_start, s_len, t_start, t_len = int(s[0]), int(s[1]), int(t[0]), int(t[1])
            s_end = s_start + s_len - 1
            interval = t_start - s_end - 1
            ext = min(int(s[5]), int(t[4])) - 2
            if interval < 0 or ext > s_len:
                continue
            min_sa = (params.pb_coverage // 60) * 3 if params.pb_coverage > 60 else 3
            descs.append((read[s_start: s_start + s_len][s_len - ext:], read[s_end + 1: s_end + 1 + interval], read[t_start: t_start + t_len],
                          interval, ext, ext + 2, min_sa))
    return descs


def _check(hw, api, oracle, small_ds, ds_units, genome, cov, tables, wide, n_reads, modes=(0, 1), skip=0):
    (u0, u1), n_sym = ds_units
    h = hw.index(u0, u1, n_sym, wide=wide, tables=tables)
    p = api.params_default(genome, cov)
    ob, orb = oracle.bwt_load(small_ds.prefix + ".bwt"), oracle.bwt_load(small_ds.prefix + ".rbwt")
    off = small_ds.off[: n_reads + 1].copy()
    bases = small_ds.bases[: int(off[-1])]
    count, seeds, _ = oracle.find_seeds(ob, orb, p, bases, off)
    descs = _skip_descs(p, small_ds.reads[:n_reads], count, seeds, skip) if skip else _walk_descs(p, small_ds.reads[:n_reads], count, seeds)
    codes, fast, steps = {}, 0, 0
    for d in descs:
        wcode, wmerged, wst = oracle.extend_walk(ob, orb, p, *d)
        for mode in modes:
            code, merged, st, nf = hw.extend_walk(h, p, *d, mode)
            assert (code, merged, st) == (wcode, wmerged, wst[0]), (mode, d)
            if mode == 1:
                fast += nf; steps += st
        codes[wcode] = codes.get(wcode, 0) + 1
    hw.index_free(h)
    ob.close(); orb.close()
    return len(descs), codes, fast, steps


def test_host_walk_matches_oracle(hw, api, oracle, small_ds, ds_units):
    n, codes, fast, steps = _check(hw, api, oracle, small_ds, ds_units, 5, 90, (5, 9, 11), False, 40)
    assert n > 100 and codes.get(1, 0) > n // 2 and codes.get(-1, 0) > 0
    assert fast > steps // 2                                  # the fast path carries most steps, the hand-over the rest


def test_host_walk_without_tables_and_wide_layout(hw, api, oracle, small_ds, ds_units):
    _check(hw, api, oracle, small_ds, ds_units, 10, 90, (), False, 12)
    _check(hw, api, oracle, small_ds, ds_units, 5, 90, (5, 9), True, 12)


def test_host_walk_long_gaps(hw, api, oracle, small_ds, ds_units):
    """Seed i to seed i + 4: result paths beyond 64 words, frontiers of many leaves, walks that fail after hundreds of steps."""
    n, codes, fast, steps = _check(hw, api, oracle, small_ds, ds_units, 5, 90, (5, 9, 11), False, 30, skip=4)
    assert n >= 40 and steps > 20 * n and sum(v for c, v in codes.items() if c <= 0) > 0


def test_host_walk_repeat_dataset(hw, api, oracle, repeat_ds):
    """Repeat-rich reads: repeat-to-unique walks on the reverse strand, the isInsufficientFreqs / SelectFreqsOfrange branch."""
    n, codes, fast, steps = _check(hw, api, oracle, repeat_ds, _units(repeat_ds), 5, 90, (5, 9), False, 60)
    assert n > 100 and len(codes) >= 2
