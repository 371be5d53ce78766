"""Golden vectors for the host-side helpers of `stride pbcorrect --onlyseed -b` and `stride kmercheck`, made from the
REFERENCE'S OWN object code: oracle/_ref/liblrsc_ref.so links PacBio/BCode.cpp and Util/KmerDistribution.cpp compiled where
they lie under /root/reference (oracle/Makefile).  Run in the build container:

    make -C oracle ref && python tests/golden/make_host_tools.py

Writes tests/golden/host_tools.json (inputs + the reference's answers) and tests/golden/barcode_sample.txt (a barcode file in
the nine-column layout BCode::load reads).  Only data is stored: no reference source text.
"""
from __future__ import annotations

import ctypes as C
import json
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
OUT = Path(__file__).resolve().parent


def make_block(rng, seq_len=320):
    # a read with homopolymer runs (so that insertion / deletion checks sometimes pass) and a block of error marks over it
    seq = []
    while len(seq) < seq_len:
        seq += [rng.choice(list("ACGT"))] * int(rng.choice([1, 1, 1, 2, 2, 3, 4]))
    seq = "".join(seq[:seq_len])
    start = int(rng.integers(0, 30))
    end = int(rng.integers(seq_len - 40, seq_len))
    n = end - start
    code = []
    i = 0
    while i < n:
        ins = "0"
        dele = "0"
        u = rng.random()
        if u < 0.03:
            ins = "1"
        elif u < 0.035:
            ins = "2"
        v = rng.random()
        if v < 0.03:
            dele = rng.choice(list("1248"))
        elif v < 0.04:
            dele = rng.choice(list("3569ac"))
        code.append(ins + dele)
        i += 1
    # a few longer insertion runs
    for _ in range(int(rng.integers(0, 4))):
        p = int(rng.integers(20, n - 20))
        for t in range(int(rng.integers(2, 4))):
            code[p + t] = "1" + code[p + t][1]
    return dict(seq=seq, start=start, end=end, rvc=int(rng.integers(0, 2)), code="".join(code))


def main():
    lib = C.CDLL(str(ROOT / "oracle" / "_ref" / "liblrsc_ref.so"))
    lib.ref_bcode_validate.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_int, C.c_char_p]
    lib.ref_bcode_validate.restype = C.c_int
    lib.ref_bcode_load_dump.argtypes = [C.c_char_p, C.c_char_p, C.c_uint64]
    lib.ref_bcode_load_dump.restype = C.c_uint64
    lib.ref_kd_compare.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.c_char_p, C.c_uint64]
    lib.ref_kd_compare.restype = C.c_uint64

    rng = np.random.default_rng(0xBC0DE)
    blocks = []
    for _ in range(48):
        b = make_block(rng)
        cases = []
        margin = 16
        for _ in range(120):
            k = int(rng.integers(9, 41))
            pos = int(rng.integers(b["start"] + margin, b["end"] - k - margin))
            cases.append([pos, k])
        # k-mers flush with the block's edges where no error mark is near (no out-of-range access in the reference there)
        n = b["end"] - b["start"]
        if set(b["code"][: 2 * 48]) == {"0"}:
            cases.append([b["start"], 21])
        if set(b["code"][2 * (n - 48):]) == {"0"}:
            cases.append([b["end"] - 21, 21])
        b["cases"] = cases
        b["verdicts"] = [lib.ref_bcode_validate(p, k, b["start"], b["end"], b["code"].encode(), b["rvc"], b["seq"].encode()) for p, k in cases]
        blocks.append(b)
    # malformed input: a digit outside 0-9a-f inside the k-mer's slice; a k-mer that starts beyond the code
    bad = make_block(rng)
    code = list(bad["code"])
    code[2 * 60] = "g"
    bad["code"] = "".join(code)
    bad["cases"] = [[bad["start"] + 50, 21], [bad["start"] + 100, 21], [bad["end"] + 10, 15]]
    bad["verdicts"] = [lib.ref_bcode_validate(p, k, bad["start"], bad["end"], bad["code"].encode(), bad["rvc"], bad["seq"].encode())
                       for p, k in bad["cases"]]
    blocks.append(bad)

    # BCode::load: nine columns, several blocks per read, no trailing newline (so that the reference's reader loop ends cleanly)
    lines = []
    for i, b in enumerate(blocks[:6]):
        lines.append(f"read{i // 2} {b['start']} {b['end']} ref{i} {1000 + i} {1300 + i} {b['code']} {'True' if b['rvc'] else 'False'} {i * 7}")
    text = "\n".join(lines)
    (OUT / "barcode_sample.txt").write_text(text)
    buf = C.create_string_buffer(1 << 20)
    n = lib.ref_bcode_load_dump(str(OUT / "barcode_sample.txt").encode(), buf, len(buf))
    load_dump = buf.raw[:n].decode()

    # KmerDistribution::compare
    kd = []
    for t in range(12):
        crt = rng.poisson(40 + 5 * t, size=int(rng.integers(30, 400))).astype(np.int32) + 2
        err = (rng.poisson(3, size=int(rng.integers(20, 300))) + 2).astype(np.int32)
        if t % 4 == 3:
            err = np.concatenate([err, rng.integers(2, 120, size=40).astype(np.int32)])
        crt = np.ascontiguousarray(crt); err = np.ascontiguousarray(err)
        n = lib.ref_kd_compare(crt.ctypes.data, crt.size, err.ctypes.data, err.size, 90, 15 + t, buf, len(buf))
        kd.append(dict(cov=90, k=15 + t, crt=crt.tolist(), err=err.tolist(), text=buf.raw[:n].decode()))

    (OUT / "host_tools.json").write_text(json.dumps(dict(
        note="answers of the reference's BCode.cpp / KmerDistribution.cpp object code (oracle/_ref), see make_host_tools.py",
        blocks=blocks, load_dump=load_dump, kd=kd)))
    tot = sum(len(b["cases"]) for b in blocks)
    ok = sum(v == 1 for b in blocks for v in b["verdicts"])
    print(f"{tot} validate cases ({ok} correct, {sum(v == 0 for b in blocks for v in b['verdicts'])} wrong, "
          f"{sum(v < 0 for b in blocks for v in b['verdicts'])} throwing); {len(kd)} compare cases")


if __name__ == "__main__":
    main()
