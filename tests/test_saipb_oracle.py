"""SURVEY section 8 row f3, oracle side only: the restatement of SAIPBSelfCorrectTree's hash-guided seed-to-seed extension
(oracle/saipb_oracle.cpp) driven like its one -- commented-out -- call site in the reference.  PARITY UNPINNED: the class is never
instantiated by the reference and cannot be built here (google dense_hash); what pins it is (a) the layers underneath
(findInterval / updateInterval / getChar / getOcc and aln_stdaln, all checked against the reference's object code) and (b) the
semantic check below: its merged paths are exact substrings of the genome the reads were simulated from far more often than the
raw reads are.  There is no device kernel for this row yet."""
from __future__ import annotations

import hashlib
import json

import numpy as np

from .conftest import GOLDEN


def _pairs(oracle, api, ds, n_reads):
    from oracle.oracle_py import unpack_reads

    ob, orb = oracle.bwt_load(ds.prefix + ".bwt"), oracle.bwt_load(ds.prefix + ".rbwt")
    p = api.params_default(5, 90)
    off = ds.off[: n_reads + 1].copy()
    bases = ds.bases[: int(off[-1])]
    count, seeds, _ = oracle.find_seeds(ob, orb, p, bases, off)
    reads = unpack_reads(bases, off)
    first = np.concatenate([[0], np.cumsum(count)]).astype(int)
    out = []
    for r in range(n_reads):
        s = seeds[first[r]: first[r + 1]].tolist()
        for j in range(1, len(s)):
            s_end, t0, t_len = s[j - 1][0] + s[j - 1][1], s[j][0], s[j][1]
            if s_end < 60 or t0 <= s_end:
                continue
            out.append((r, reads[r][s_end - 60: s_end], reads[r][s_end: t0], reads[r][t0: t0 + t_len], t0 - s_end))
    return ob, orb, reads, out


def test_saipb_oracle_merges_seed_pairs_into_genome_substrings(api, oracle, small_ds):
    ob, orb, reads, pairs = _pairs(oracle, api, small_ds, 60)
    genome = small_ds.genome.tobytes().decode()
    rc_genome = genome[::-1].translate(str.maketrans("ACGT", "TGCA"))
    codes, exact, raw_exact, digest = {}, 0, 0, hashlib.sha256()
    for r, source, between, target, dis in pairs:
        code, merged, st = oracle.saipb_merge(ob, orb, source, between, target, dis)
        codes[code] = codes.get(code, 0) + 1
        digest.update(f"{code}:{merged}\n".encode())
        if code != 1:
            assert merged == "" and code in (-1, -2, -3, -4, -5)
            continue
        assert merged.startswith(source) and merged.endswith(target) and st["results"] >= 1
        mid = merged[len(source) - 17:]
        exact += (mid in genome) or (mid in rc_genome)
        raw = source[-17:] + between + target
        raw_exact += (raw in genome) or (raw in rc_genome)
    ok = codes.get(1, 0)
    assert len(pairs) > 500 and ok > 250
    assert exact > 0.7 * ok and raw_exact < 0.05 * ok          # 76 % of the merged paths are error-free, 1 % of the raw segments
    # regression pin of the restatement's own output (not a reference pin)
    gold = json.loads((GOLDEN / "saipb_oracle.json").read_text())
    assert {str(k): v for k, v in codes.items()} == gold["codes"] and digest.hexdigest() == gold["sha256"]
    ob.close(); orb.close()
