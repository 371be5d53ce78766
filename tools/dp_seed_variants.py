"""On-GPU experiment for the `optnone` on dp_seed_kernel: runs the dp-consensus parity case with every compiled
variant of the kernel (LRSC_DP_SEED_VARIANT, dp_retrieve.hip) and reports which ones match the CPU oracle."""
import os, sys, tempfile
from pathlib import Path
import numpy as np
REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))
from longreadselfcorrect_amd import Lrsc
from oracle import oracle_py
from tests.conftest import Dataset
from tests.test_gpu_fm import _dp_queries

api, orc = Lrsc(), oracle_py.Oracle()
orc._decl_late()
with tempfile.TemporaryDirectory() as tmp:
    ds = Dataset(api, orc, tmp, 4000, 180, 2000)
    idx = api.index_open(ds.prefix + ".bwt", ds.prefix + ".rbwt"); idx.upload(0)
    ob, orb = orc.bwt_load(ds.prefix + ".bwt"), orc.bwt_load(ds.prefix + ".rbwt")
    for cov in (90, 20):
        rng = np.random.default_rng(1234 + cov)
        qs = _dp_queries(rng, ds, 60)
        want = [orc.dp_consensus(ob, orb, q, k, mo, mi, cov, mc) for (q, k, mo, mi, mc) in qs]
        for v in range(6):
            os.environ["LRSC_DP_SEED_VARIANT"] = str(v)
            ctx = idx.ctx(api.params_default(5, cov), 0)
            got = ctx.dp_consensus(qs)
            ctx.close()
            bad = [i for i, (g, w) in enumerate(zip(got, want)) if g != w]
            print(f"cov {cov} variant {v}: {'PASS' if not bad else 'FAIL'} ({len(bad)} of {len(qs)} differ; first {bad[:5]}; ks {[qs[i][1] for i in bad[:5]]})", flush=True)
