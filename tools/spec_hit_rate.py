"""How often is the source k-mer of a seed-to-seed walk known before the previous walk has run?

The walk-parallel schedule assumes that walk j of a read starts from the last k characters of
seed j-1's OWN string (the read substring), which is what the accumulated source string ends with after an FM success
that terminates at target offset 0 (LongReadCorrectByOverlap.cpp:849-851), after the raw-copy fallback
(PacBioSelfCorrectionProcess.cpp:146) and after --split (:143).  This tool measures, on the CPU oracle, how often that
holds (CorrectionResult::spec in oracle/process_oracle.hpp).  CPU only; uses oracle/ => a tool, not product code.

    python tools/spec_hit_rate.py [--long-reads 48]
"""
import argparse
import json
import sys
import tempfile
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))


def run(api, orc, name, genome_len, n_reads, tmpl_len, seed, n_correct, variants, out, genome=None):
    g = api.synth_genome(seed, genome_len) if genome is None else genome
    bases, off = api.synth_reads(seed + 1, g, n_reads, tmpl_len)
    with tempfile.TemporaryDirectory() as d:
        orc.build_index(bases, off, d + "/x")
        ob, orb = orc.bwt_load(d + "/x.bwt"), orc.bwt_load(d + "/x.rbwt")
        sub_off = off[: n_correct + 1].copy()
        sub = bases[: int(sub_off[-1])]
        for vname, kw in variants:
            p = api.params_default(5, 90)
            for k, v in kw.items():
                setattr(p, k, v)
            r = orc.correct_reads(ob, orb, p, sub, sub_off)
            s = r.spec_stats
            s["hit_rate"] = round(s["hit"] / max(1, s["walks"]), 4)
            out[f"{name}/{vname}"] = s
            print(name, vname, s, flush=True)
            r.close()
        ob.close(); orb.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--long-reads", type=int, default=48)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    from longreadselfcorrect_amd import Lrsc
    from oracle import oracle_py

    oracle_py.build_oracle()
    api, orc = Lrsc(), oracle_py.Oracle()
    out = {}
    V = [("default", {}), ("nodp", {"no_dp": 1}), ("split", {"split": 1})]
    run(api, orc, "small_2kb", 4000, 180, 2000, 0x5EED0001, 180, V, out)
    # the repeat-rich set of tests/conftest.py (isRepeat seeds, repeat-to-unique walks)
    g = api.synth_genome(0xBEEF, 14000).copy()
    unit = g[50:110].copy()
    for c in range(75):
        g[4000 + c * 80: 4000 + c * 80 + 60] = unit
    seg1 = g[300:650].copy()
    for pos in (1200, 2100, 3000, 10500, 12000):
        g[pos: pos + 350] = seg1
    run(api, orc, "repeat_2kb", 14000, 630, 2000, 0xBEEF, 200, V[:2], out, genome=g)
    run(api, orc, "long_10kb", 20000, 180, 10000, 0x10C0FFEE, a.long_reads, V[:2], out)
    if a.out:
        Path(a.out).write_text(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
