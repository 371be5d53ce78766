"""Host-side helpers of `stride pbcorrect --onlyseed -b FILE` and `stride kmercheck` (longreadselfcorrect_amd/host/BCode.cpp,
KmerDistribution.h) against golden vectors made from the reference's own object code (tests/golden/make_host_tools.py), and --
where oracle/_ref is built -- against that object code live on fresh random blocks."""
from __future__ import annotations

import ctypes as C
import json
import subprocess

import numpy as np
import pytest

from .conftest import GOLDEN, REPO

GOLD = json.loads((GOLDEN / "host_tools.json").read_text())


@pytest.fixture(scope="module")
def driver(tmp_path_factory):
    exe = tmp_path_factory.mktemp("host_tools") / "driver"
    subprocess.run(["g++", "-std=c++14", "-O2", "-Wall", "-Wextra", "-Werror", "-o", str(exe), str(REPO / "tests/host_tools/driver.cpp"),
                    str(REPO / "longreadselfcorrect_amd/host/BCode.cpp"), "-lz"], check=True)
    return str(exe)


def _validate(driver, blocks):
    text = "".join(f"{p} {k} {b['start']} {b['end']} {b['rvc']} {b['code']} {b['seq']}\n" for b in blocks for p, k in b["cases"])
    out = subprocess.run([driver, "validate"], input=text, capture_output=True, text=True, check=True).stdout.split()
    return [int(x) for x in out]


def test_bcode_validate_matches_the_reference_object_code(driver):
    got = _validate(driver, GOLD["blocks"])
    want = [v for b in GOLD["blocks"] for v in b["verdicts"]]
    assert len(got) == len(want) > 5000
    assert got == want
    assert want.count(1) > 1000 and want.count(0) > 1000 and want.count(-1) == 2      # correct, wrong and throwing k-mers all occur


def test_bcode_load_reads_the_nine_column_file(driver):
    out = subprocess.run([driver, "load", str(GOLDEN / "barcode_sample.txt")], capture_output=True, text=True, check=True)
    assert out.stdout == GOLD["load_dump"]
    assert "Loading BARCODE: " in out.stderr


def test_bcode_load_accepts_a_trailing_newline_and_gzip(driver, tmp_path):
    import gzip

    text = (GOLDEN / "barcode_sample.txt").read_text() + "\n"
    p = tmp_path / "b.txt.gz"
    with gzip.open(p, "wt") as f:
        f.write(text)
    out = subprocess.run([driver, "load", str(p)], capture_output=True, text=True, check=True)
    assert out.stdout == GOLD["load_dump"]


def test_kmer_distribution_compare(driver):
    for case in GOLD["kd"]:
        text = " ".join(map(str, case["crt"])) + "\n" + " ".join(map(str, case["err"])) + "\n"
        out = subprocess.run([driver, "compare", str(case["cov"]), str(case["k"])], input=text, capture_output=True, text=True, check=True)
        assert out.stdout == case["text"]


def test_bcode_validate_fuzz_against_live_reference(driver, ref):
    """Fresh random blocks every run of the generator seed below, checked against oracle/_ref (skipped where it is not built)."""
    from tests.golden.make_host_tools import make_block

    lib = ref.lib
    lib.ref_bcode_validate.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_int, C.c_char_p]
    lib.ref_bcode_validate.restype = C.c_int
    rng = np.random.default_rng(20261004)
    blocks = []
    for _ in range(30):
        b = make_block(rng)
        b["cases"] = []
        for _ in range(150):
            k = int(rng.integers(9, 51))
            b["cases"].append([int(rng.integers(b["start"] + 16, b["end"] - k - 16)), k])
        blocks.append(b)
    want = [lib.ref_bcode_validate(p, k, b["start"], b["end"], b["code"].encode(), b["rvc"], b["seq"].encode())
            for b in blocks for p, k in b["cases"]]
    assert _validate(driver, blocks) == want
