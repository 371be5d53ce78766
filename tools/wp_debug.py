"""Debug driver (GPU box): the walk-parallel flow against the oracle on the small test set, per-read status and counters."""
import os, sys, tempfile
from pathlib import Path
import numpy as np
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
from longreadselfcorrect_amd import Lrsc
from oracle import oracle_py

nodp = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
api, orc = Lrsc(), oracle_py.Oracle()
genome = api.synth_genome(0x5EED0001, 4000)
bases, off = api.synth_reads(0x5EED0002, genome, 180, 2000)
with tempfile.TemporaryDirectory() as d:
    orc.build_index(bases, off, d + "/x")
    ob, orb = orc.bwt_load(d + "/x.bwt"), orc.bwt_load(d + "/x.rbwt")
    index = api.index_open(d + "/x.bwt", d + "/x.rbwt") if hasattr(api, "index_open") else None
    p = api.params_default(5, 90)
    p.no_dp = nodp
    sub_off = off[: n + 1].copy()
    sub = bases[: int(sub_off[-1])]
    want = orc.correct_reads(ob, orb, p, sub, sub_off)
    index.upload(0)
    ctx = index.ctx(p, 0)
    results, pieces = ctx.correct_reads(sub, sub_off)
    wc = want.counters
    for i, r in enumerate(results):
        got = [r.total_reads_len, r.corrected_len, r.total_seed_num, r.total_walk_num, r.high_error_num, r.exceed_depth_num,
               r.exceed_leave_num, r.fm_num, r.dp_num, r.seed_dis, r.merge]
        ok = got == list(wc[i])
        print(i, "status", r.status, "OK" if ok else f"DIFF got {got} want {list(wc[i])}")
    got_fa = "".join(f">r{i}\n{p_[0]}\n" for i, (r, p_) in enumerate(zip(results, pieces)) if r.merge)
    print("fasta identical:", got_fa == want.correct_fa)
