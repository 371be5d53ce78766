"""Shared fixtures.  GPU tests are marked @pytest.mark.gpu; everything else runs on CPU."""
from __future__ import annotations

import sys
from pathlib import Path

import numpy as np
import pytest

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))

GOLDEN = REPO / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def api():
    """The product C-ABI library (must be built: no fallback)."""
    import __graft_entry__ as g

    g.build_product()
    from longreadselfcorrect_amd import Lrsc

    return Lrsc()


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle_py

    oracle_py.build_oracle()
    return oracle_py.Oracle()


@pytest.fixture(scope="session")
def ref():
    from oracle import oracle_py

    try:
        ok = oracle_py.build_ref()
    except Exception:  # reference tree present but failed to compile: surface it
        raise
    if not ok:
        pytest.skip("oracle/_ref not built (no /root/reference here); golden fixtures cover this")
    return oracle_py.Ref()


class Dataset:
    """A small self-correction workload: reads simulated from a random genome at high coverage,
    indexed over themselves."""

    def __init__(self, api, oracle, tmp, genome_len, n_reads, tmpl_len, seed=0x5EED0001):
        self.genome = api.synth_genome(seed, genome_len)
        self.bases, self.off = api.synth_reads(seed + 1, self.genome, n_reads, tmpl_len)
        self.prefix = str(Path(tmp) / "reads")
        oracle.build_index(self.bases, self.off, self.prefix)
        self.n_reads = n_reads

    @property
    def reads(self):
        from oracle.oracle_py import unpack_reads

        return unpack_reads(self.bases, self.off)


@pytest.fixture(scope="session")
def small_ds(api, oracle, tmp_path_factory):
    # 4 kb genome, 180 x 2 kb templates = 90x: ~0.4 M symbols per strand
    return Dataset(api, oracle, tmp_path_factory.mktemp("small_ds"), 4000, 180, 2000)


def write_fasta(path, reads, prefix="r"):
    with open(path, "w") as f:
        for i, r in enumerate(reads):
            f.write(f">{prefix}{i}\n{r}\n")


class RepeatDataset(Dataset):
    """Like Dataset, but the genome carries a 75-copy interspersed repeat (60-bp unit + 20-bp unique spacers) and
    a 6-copy 350-bp repeat, so that repeat-mode attributes (19-mers above the ~640x repeat threshold at 90x),
    isRepeat seeds and repeat-to-unique (reverse strand) walks occur."""

    def __init__(self, api, oracle, tmp, seed=0xBEEF):
        g = api.synth_genome(seed, 14000).copy()
        unit = g[50:110].copy()
        for c in range(75):
            pos = 4000 + c * 80
            g[pos: pos + 60] = unit
        seg1 = g[300:650].copy()
        for pos in (1200, 2100, 3000, 10500, 12000):
            g[pos: pos + 350] = seg1
        self.genome = g
        self.n_reads = 630
        self.bases, self.off = api.synth_reads(seed + 1, g, self.n_reads, 2000)
        self.prefix = str(Path(tmp) / "rep")
        oracle.build_index(self.bases, self.off, self.prefix)


@pytest.fixture(scope="session")
def repeat_ds(api, oracle, tmp_path_factory):
    return RepeatDataset(api, oracle, tmp_path_factory.mktemp("repeat_ds"))
