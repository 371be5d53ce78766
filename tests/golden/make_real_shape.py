"""Regenerates tests/golden/real_shape.json: the CPU oracle's outputs on the two bench-shaped datasets of
tests/test_gpu_real_shape.py (10 kb reads; an index large enough for the 13/15-mer tables).  Runs on the CPU in the
build container (minutes):  python tests/golden/make_real_shape.py

The datasets come from the repo's own deterministic generator; only digests and counter sums are stored.
"""
import hashlib
import json
import sys
import tempfile
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(REPO))
OUT = Path(__file__).resolve().parent

LONG = dict(genome_len=20_000, n_reads=180, tmpl_len=10_000, seed=0x10C0FFEE, correct_reads=64)
BIG = dict(genome_len=790_000, n_reads=6_800, tmpl_len=10_000, seed=0xB16B00B5, seed_reads=200, correct_reads=32)


def digest(s) -> str:
    return hashlib.sha256(s if isinstance(s, bytes) else s.encode()).hexdigest()


def make(api, which):
    genome = api.synth_genome(which["seed"], which["genome_len"])
    bases, off = api.synth_reads(which["seed"] + 1, genome, which["n_reads"], which["tmpl_len"])
    return bases, off


def run_correct(api, orc, ob, orb, bases, off, n, nodp):
    p = api.params_default(5, 90)
    p.no_dp = nodp
    sub_off = off[: n + 1].copy()
    run = orc.correct_reads(ob, orb, p, bases[: int(sub_off[-1])], sub_off)
    out = dict(reads=n, no_dp=nodp, correct_fa_sha256=digest(run.correct_fa), correct_fa_bytes=len(run.correct_fa),
               discard_fa_sha256=digest(run.discard_fa), counter_sums=run.counters.sum(axis=0).tolist())
    run.close()
    return out


def main():
    from longreadselfcorrect_amd import Lrsc
    from oracle import oracle_py

    api, orc = Lrsc(), oracle_py.Oracle()
    out = {"long": dict(LONG), "big": dict(BIG)}
    with tempfile.TemporaryDirectory() as d:
        bases, off = make(api, LONG)
        orc.build_index(bases, off, d + "/long")
        ob, orb = orc.bwt_load(d + "/long.bwt"), orc.bwt_load(d + "/long.rbwt")
        out["long"]["num_symbols"] = int(ob.num_symbols)
        out["long"]["bwt_sha256"] = digest(open(d + "/long.bwt", "rb").read())
        out["long"]["default"] = run_correct(api, orc, ob, orb, bases, off, LONG["correct_reads"], 0)
        out["long"]["nodp"] = run_correct(api, orc, ob, orb, bases, off, LONG["correct_reads"], 1)
        ob.close(); orb.close()
        print("long done", flush=True)

        bases, off = make(api, BIG)
        orc.build_index(bases, off, d + "/big")
        ob, orb = orc.bwt_load(d + "/big.bwt"), orc.bwt_load(d + "/big.rbwt")
        out["big"]["num_symbols"] = int(ob.num_symbols)
        out["big"]["bwt_sha256"] = digest(open(d + "/big.bwt", "rb").read())
        out["big"]["rbwt_sha256"] = digest(open(d + "/big.rbwt", "rb").read())
        n = BIG["seed_reads"]
        sub_off = off[: n + 1].copy()
        count, seeds, attr = orc.find_seeds(ob, orb, api.params_default(5, 90), bases[: int(sub_off[-1])], sub_off)
        out["big"]["seeds"] = dict(reads=n, n_seeds=int(count.sum()), count_sha256=digest(np.ascontiguousarray(count, dtype=np.uint32).tobytes()),
                                   seeds_sha256=digest(np.ascontiguousarray(seeds, dtype=np.int32).tobytes()),
                                   attribute_sha256=digest(np.ascontiguousarray(attr, dtype=np.int8).tobytes()))
        out["big"]["default"] = run_correct(api, orc, ob, orb, bases, off, BIG["correct_reads"], 0)
        ob.close(); orb.close()
    (OUT / "real_shape.json").write_text(json.dumps(out, indent=1))
    print("written", OUT / "real_shape.json")


if __name__ == "__main__":
    main()
