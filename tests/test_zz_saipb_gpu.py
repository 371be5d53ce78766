"""Row f3 on the device's FM primitives (`-m gpu`; named to run last): the host SAIPBSelfCorrectTree with its FM access over the
C ABI -- lrsc_find_kmers for the k-mer intervals, one lrsc_rank batch per step for the extensions of all leaves, lrsc_lf_walk for
the k-mer collection -- against the oracle restatement on the same seed pairs (FM-walk code and merged sequence identical)."""
from __future__ import annotations

import pytest

from .test_saipb_host import build_driver, run_pairs
from .test_saipb_oracle import _pairs

pytestmark = pytest.mark.gpu


def test_host_tree_over_the_device_fm_primitives_matches_oracle(api, oracle, small_ds, tmp_path):
    exe = build_driver(tmp_path, False)
    ob, orb, _, pairs = _pairs(oracle, api, small_ds, 60)          # the full seed-pair set of the CPU test (> 500 pairs)
    assert len(pairs) > 500
    got = run_pairs(exe, "device", small_ds, pairs)
    want = [oracle.saipb_merge(ob, orb, s, b, t, d)[:2] for _, s, b, t, d in pairs]
    assert got == want
    assert sum(c == 1 for c, _ in want) > 250 and sum(c < 0 for c, _ in want) > 20
    ob.close(); orb.close()
