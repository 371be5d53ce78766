"""GPU index construction (row f1: `stride index`) vs the oracle builder, which is itself pinned
byte-for-byte to the reference's ropebwt2 output (tests/test_oracle_vs_ref.py)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _file_units(path):
    raw = np.fromfile(path, dtype=np.uint8)
    return raw[30:]


@pytest.mark.parametrize("reverse,ext", [(False, "bwt"), (True, "rbwt")])
def test_gpu_bwt_equals_oracle_bytes(api, small_ds, reverse, ext):
    units = api.build_bwt(small_ds.bases, small_ds.off, reverse, 0)
    np.testing.assert_array_equal(units, _file_units(f"{small_ds.prefix}.{ext}"))


def test_gpu_bwt_file_roundtrip(api, small_ds, tmp_path):
    units = api.build_bwt(small_ds.bases, small_ds.off, False, 0)
    n_sym = int(small_ds.off[-1]) + small_ds.n_reads
    api.write_bwt_file(tmp_path / "x.bwt", units, small_ds.n_reads, n_sym)
    assert (tmp_path / "x.bwt").read_bytes() == open(small_ds.prefix + ".bwt", "rb").read()


def test_gpu_bwt_pathological_ties(api, oracle, tmp_path):
    """Identical reads, reads that are prefixes/suffixes of each other, homopolymers and 1-base reads:
    sentinel ties must follow input order (MR_SO_IO) and long LCPs need many refinement rounds."""
    from oracle.oracle_py import pack_reads

    rng = np.random.default_rng(3)
    base = "".join(rng.choice(list("ACGT"), size=300))
    reads = [base, base, base[:150], base[150:], "A" * 200, "A" * 199, "A", "C", base[::-1], base, "ACGT" * 40, "T"]
    bases, off = pack_reads(reads)
    oracle.build_index(bases, off, str(tmp_path / "p"))
    for reverse, ext in [(False, "bwt"), (True, "rbwt")]:
        units = api.build_bwt(bases, off, reverse, 0)
        np.testing.assert_array_equal(units, _file_units(tmp_path / f"p.{ext}"))


@pytest.mark.parametrize("job,wide_pos", [(20000, 0), (20000, 1), (0, 1), (3000, 1)])
def test_gpu_bwt_grouped_jobs_and_64bit_positions(api, oracle, small_ds, tmp_path, monkeypatch, job, wide_pos):
    """The path of read sets above the per-job limit (and above 2^32 symbols), forced at small size: suffixes cut into groups of
    leading-symbol classes that are sorted one after the other (LRSC_BWT_JOB = suffixes per job), 64-bit text positions
    (LRSC_BWT_WIDE_POS).  Byte-identical to the one-job build, on the regular and on the pathological read set."""
    from oracle.oracle_py import pack_reads

    if job:
        monkeypatch.setenv("LRSC_BWT_JOB", str(job))
    monkeypatch.setenv("LRSC_BWT_WIDE_POS", str(wide_pos))
    for reverse, ext in [(False, "bwt"), (True, "rbwt")]:
        units = api.build_bwt(small_ds.bases, small_ds.off, reverse, 0)
        np.testing.assert_array_equal(units, _file_units(f"{small_ds.prefix}.{ext}"))
    rng = np.random.default_rng(3)
    base = "".join(rng.choice(list("ACGT"), size=300))
    reads = [base, base, base[:150], base[150:], "A" * 200, "A" * 199, "A", "C", base[::-1], base, "ACGT" * 40, "T"]
    bases, off = pack_reads(reads)
    oracle.build_index(bases, off, str(tmp_path / "p"))
    for reverse, ext in [(False, "bwt"), (True, "rbwt")]:
        units = api.build_bwt(bases, off, reverse, 0)
        np.testing.assert_array_equal(units, _file_units(tmp_path / f"p.{ext}"))
