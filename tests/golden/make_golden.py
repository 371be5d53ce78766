"""Regenerates the fixtures under tests/golden/ from the reference's own object code
(oracle/_ref/liblrsc_ref.so, built by oracle/Makefile from /root/reference).  Run in the build
container only:  python tests/golden/make_golden.py

Fixtures are data (inputs + expected outputs); no reference source text is stored.
"""
import json
import subprocess
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(REPO))
OUT = Path(__file__).resolve().parent


def threshold_tables():
    # KmerThreshold is an initialise-once singleton: one process per coverage
    tables = {}
    for cov in (20, 30, 60, 90, 120, 200):
        code = ("import sys; sys.path.insert(0, %r); from oracle import oracle_py as o;"
                "import json; print(json.dumps(o.Ref().threshold_table(%d).tolist()))" % (str(REPO), cov))
        out = subprocess.run([sys.executable, "-c", code], check=True, capture_output=True, text=True).stdout
        tables[str(cov)] = json.loads(out)
    (OUT / "threshold_tables.json").write_text(json.dumps(tables))


def small_dataset(api, orc, tmp):
    from tests.conftest import Dataset

    return Dataset(api, orc, tmp, 4000, 180, 2000)          # == the small_ds fixture of tests/conftest.py


def reference_kats(ds, ref):
    """Known answers from the reference's own object code (oracle/_ref): ropebwt2 index bytes, RLBWT::getOcc / getChar,
    Overlapper::extendMatch -- on the seeded small dataset of the test suite."""
    import hashlib
    import tempfile

    import numpy as np

    from tests.conftest import write_fasta

    out = {}
    with tempfile.TemporaryDirectory() as d:
        fa = Path(d) / "reads.fa"
        write_fasta(fa, ds.reads)
        ref.build_index(str(fa), str(Path(d) / "ref"))
        for ext in ("bwt", "rbwt"):
            raw = open(f"{d}/ref.{ext}", "rb").read()
            out[f"{ext}_sha256"] = hashlib.sha256(raw).hexdigest()
            out[f"{ext}_bytes"] = len(raw)
        rb = ref.bwt_load(f"{d}/ref.bwt")
        rng = np.random.default_rng(20260101)
        n = rb.num_symbols
        idx = np.concatenate([np.array([-1, 0, 1, 31, 32, 127, 128, 191, 192, n - 2, n - 1]), rng.integers(-1, n, 500)]).astype(np.int64)
        base = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, idx.size)]
        out["occ"] = {"idx": idx.tolist(), "base": bytes(base).decode(), "occ": rb.occ(base, idx).tolist()}
        rows = rng.integers(0, n, 300).astype(np.uint64)
        out["chars"] = {"rows": rows.tolist(), "chars": bytes(rb.chars(rows)).decode()}
        out["pc"] = {b: rb.pc(b) for b in "$ACGT"}
        rb.close()
    g = ds.genome.tobytes().decode()
    pairs = []
    for t in range(24):
        L = 60 + 13 * t
        p = 37 * t
        q = g[p: p + L]
        s2 = list(g[p: p + L + 20])
        for j in range(5 + t % 7, len(s2), 11 + t % 5):                  # deterministic substitutions / deletions / insertions
            if (j + t) % 3 == 0:
                s2[j] = "ACGT"[(("ACGT".index(s2[j])) + 1) % 4]
            elif (j + t) % 3 == 1:
                s2[j] = ""
            else:
                s2[j] = s2[j] + "ACGT"[(j + t) % 4]
        s2 = "".join(s2)
        if t % 2:
            pairs.append((q, s2[:17] and (s2[-len(q) - 5:-17] + q[-17:]), len(q) - 17, len(s2[-len(q) - 5:-17]) ))
        else:
            pairs.append((q, q[:17] + s2[17:], 0, 0))
    out["extend_match"] = [dict(s1=a, s2=b, start1=c, start2=d, want=ref.extend_match(a, b, c, d)) for a, b, c, d in pairs]
    (OUT / "reference_kats.json").write_text(json.dumps(out))


def whole_path(ds, api, orc):
    """The oracle's correct.fa / discard.fa / counters on the small dataset (the layers above the FM-index cannot be built
    from the reference here): a regression pin for the oracle and the expected output of the product."""
    import hashlib

    out = {}
    ob, orb = orc.bwt_load(ds.prefix + ".bwt"), orc.bwt_load(ds.prefix + ".rbwt")
    for name, (genome, nodp, split) in {"g5_default": (5, 0, 0), "g5_nodp": (5, 1, 0), "g5_nodp_split": (5, 1, 1), "g10_default": (10, 0, 0)}.items():
        p = api.params_default(genome, 90)
        p.no_dp, p.split = nodp, split
        run = orc.correct_reads(ob, orb, p, ds.bases, ds.off)
        out[name] = dict(genome=genome, no_dp=nodp, split=split,
                         correct_fa_sha256=hashlib.sha256(run.correct_fa.encode()).hexdigest(), correct_fa_bytes=len(run.correct_fa),
                         discard_fa_sha256=hashlib.sha256(run.discard_fa.encode()).hexdigest(),
                         counter_sums=run.counters.sum(axis=0).tolist(), stats=run.stats)
        run.close()
    (OUT / "whole_path.json").write_text(json.dumps(out, indent=1))


if __name__ == "__main__":
    import tempfile

    from longreadselfcorrect_amd import Lrsc
    from oracle import oracle_py

    assert oracle_py.build_ref(), "needs /root/reference"
    threshold_tables()
    api, orc = Lrsc(), oracle_py.Oracle()
    with tempfile.TemporaryDirectory() as tmp:
        ds = small_dataset(api, orc, tmp)
        reference_kats(ds, oracle_py.Ref())
        whole_path(ds, api, orc)
    print("golden fixtures written to", OUT)
