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


if __name__ == "__main__":
    from oracle import oracle_py

    assert oracle_py.build_ref(), "needs /root/reference"
    threshold_tables()
    print("golden fixtures written to", OUT)
