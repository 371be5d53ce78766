"""Known answers of the reference's own stdaln object code (oracle/_ref: Thirdparty/stdaln.c compiled in place) for the call
SAIPBSelfCTree.cpp:186-194 makes: aln_stdaln(s1, s2, &aln_param_pacbio, ALN_TYPE_GLOBAL, 1) -> '|' count, score, path_len.

    make -C oracle ref && python tests/golden/make_stdaln_kats.py        -> tests/golden/stdaln_kats.json
"""
from __future__ import annotations

import ctypes as C
import json
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]


def make_pairs(rng, n):
    def mutate(s, rate):
        out = []
        for c in s:
            u = rng.random()
            if u < rate * 0.3:
                continue
            if u < rate * 0.4:
                out.append(rng.choice(list("ACGT")))
                continue
            out.append(c)
            if rng.random() < rate * 0.6:
                out.append(rng.choice(list("ACGT")))
        return "".join(out) or "A"

    pairs = []
    for t in range(n):
        L = int(rng.choice([1, 2, 3, 5, 10, 30, 60, 120, 300, 700]))
        a = "".join(rng.choice(list("ACGT"), size=L))
        kind = t % 4
        if kind == 0:
            b = mutate(a, 0.15)
        elif kind == 1:
            b = mutate(a, 0.4)
        elif kind == 2:
            b = "".join(rng.choice(list("ACGT"), size=max(1, int(L * rng.uniform(0.3, 2.5)))))
        else:
            b = a[int(L * 0.2):] + "".join(rng.choice(list("ACGT"), size=int(rng.integers(0, 80))))
        pairs.append((a, b or "C"))
    return pairs


def main():
    lib = C.CDLL(str(ROOT / "oracle" / "_ref" / "liblrsc_ref.so"))
    rng = np.random.default_rng(0x57DA1)
    cases = []
    for a, b in make_pairs(rng, 160):
        out = (C.c_int * 3)()
        lib.ref_stdaln_global(a.encode(), b.encode(), out)
        cases.append(dict(s1=a, s2=b, matches=out[0], score=out[1], path_len=out[2]))
    (Path(__file__).resolve().parent / "stdaln_kats.json").write_text(json.dumps(dict(
        note="answers of the reference's stdaln.c object code (oracle/_ref), see make_stdaln_kats.py", cases=cases)))
    print(len(cases), "cases")


if __name__ == "__main__":
    main()
