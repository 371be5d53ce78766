"""GPU debug: state-machine kernel vs the lane kernel vs the oracle on the small dataset."""
import os, sys, tempfile
from pathlib import Path
import numpy as np
REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))
from longreadselfcorrect_amd import Lrsc
from oracle import oracle_py
from tests.conftest import Dataset

api, orc = Lrsc(), oracle_py.Oracle()
names = ("total_reads_len", "corrected_len", "total_seed_num", "total_walk_num", "high_error_num", "exceed_depth_num",
         "exceed_leave_num", "fm_num", "dp_num", "seed_dis", "merge")
with tempfile.TemporaryDirectory() as tmp:
    ds = Dataset(api, orc, tmp, 4000, 180, 2000)
    idx = api.index_open(ds.prefix + ".bwt", ds.prefix + ".rbwt"); idx.upload(0)
    ob, orb = orc.bwt_load(ds.prefix + ".bwt"), orc.bwt_load(ds.prefix + ".rbwt")
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 70
    off = ds.off[: n + 1].copy(); bases = ds.bases[: int(off[-1])]
    for nodp in (1, 0):
        p = api.params_default(5, 90); p.no_dp = nodp
        want = orc.correct_reads(ob, orb, p, bases, off).counters
        for kern, rpw, ex in (("lane", None, 0), ("sm", None, 0), ("sm", None, 1), ("sm", "64", 1)):
            os.environ["LRSC_CORRECT_KERNEL"] = kern
            if ex: os.environ["LRSC_SM_EX"] = "1"
            else: os.environ.pop("LRSC_SM_EX", None)
            if rpw: os.environ["LRSC_READS_PER_WAVE"] = rpw
            else: os.environ.pop("LRSC_READS_PER_WAVE", None)
            ctx = idx.ctx(p, 0)
            try:
                res, pieces = ctx.correct_reads(bases, off)
                got = np.array([[getattr(r, f) for f in names] for r in res], dtype=np.int64)
                bad = np.flatnonzero((got != want).any(axis=1))
                print(f"nodp={nodp} kernel={kern} rpw={rpw} ex={ex}: {len(bad)} of {n} reads differ; first {bad[:8].tolist()}", flush=True)
                for b in bad[:3]:
                    print("   got ", got[b].tolist(), "\n   want", want[b].tolist(), flush=True)
            except Exception as e:
                print(f"nodp={nodp} kernel={kern} rpw={rpw} ex={ex}: ERROR {e}", flush=True)
            ctx.close()
