"""Pins the CPU restatement (oracle/) against the reference's own object code (oracle/_ref)."""
import numpy as np
import pytest

from .conftest import write_fasta


@pytest.fixture(scope="module")
def ref_index(ref, small_ds, tmp_path_factory):
    d = tmp_path_factory.mktemp("ref_idx")
    fa = d / "reads.fa"
    write_fasta(fa, small_ds.reads)
    prefix = str(d / "reads")
    ref.build_index(fa, prefix, threads=2)
    return prefix


@pytest.mark.parametrize("ext", ["bwt", "rbwt"])
def test_builder_matches_ropebwt2_bytes(ref_index, small_ds, ext):
    """Our direct suffix sort == `stride index -a ropebwt2` output, byte for byte (header + RL units)."""
    a = open(f"{ref_index}.{ext}", "rb").read()
    b = open(f"{small_ds.prefix}.{ext}", "rb").read()
    assert len(a) == len(b)
    assert a == b


@pytest.mark.parametrize("ext", ["bwt", "rbwt"])
def test_rank_and_char_match_reference(ref, oracle, small_ds, ext):
    rb = ref.bwt_load(f"{small_ds.prefix}.{ext}")
    ob = oracle.bwt_load(f"{small_ds.prefix}.{ext}")
    n = ob.num_symbols
    assert (rb.num_strings, rb.num_symbols, rb.num_runs) == (ob.num_strings, n, ob.num_runs)
    for c in "$ACGT":
        assert rb.pc(c) == ob.pc(c)
    rng = np.random.default_rng(7)
    idx = np.concatenate([
        np.array([-1, 0, 1, 15, 16, 17, 31, 32, 33, 8190, 8191, 8192, 8193, n - 2, n - 1], dtype=np.int64),
        rng.integers(-1, n, size=200_000),
    ])
    bases = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=idx.size)
    np.testing.assert_array_equal(rb.occ(bases, idx), ob.occ(bases, idx))
    pos = np.concatenate([np.arange(0, min(n, 5000)), rng.integers(0, n, size=100_000)]).astype(np.uint64)
    np.testing.assert_array_equal(rb.chars(pos), ob.chars(pos))
    rb.close(); ob.close()


def test_stdaln_global_matches_reference_object_code(ref, oracle):
    """aln_stdaln(s1, s2, &aln_param_pacbio, GLOBAL, 1) (SAIPBSelfCTree.cpp:186-194): '|' count, score and path length of the
    restatement (oracle/stdaln_oracle.cpp) against Thirdparty/stdaln.c compiled in place, on fresh random pairs."""
    import ctypes as C

    from tests.golden.make_stdaln_kats import make_pairs

    rng = np.random.default_rng(20261005)
    for a, b in make_pairs(rng, 600):
        out = (C.c_int * 3)()
        ref.lib.ref_stdaln_global(a.encode(), b.encode(), out)
        assert oracle.stdaln_global(a, b) == (out[0], out[1], out[2]), (len(a), len(b))
