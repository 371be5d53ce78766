"""Times lrsc_correct_reads (whole --nodp path) on a synthetic 90x read set; prints stage stats."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
from longreadselfcorrect_amd import Lrsc
from longreadselfcorrect_amd.capi import K_GRID, K_SEEDS, K_EXTEND, K_LF, K_DP, K_MSA

genome_mb, n_reads = float(sys.argv[1]), int(sys.argv[2])
n_correct = int(sys.argv[3]) if len(sys.argv) > 3 else n_reads
no_dp = int(sys.argv[4]) if len(sys.argv) > 4 else 1
api = Lrsc()
g = api.synth_genome(0x5EED0001, int(genome_mb * 1e6))
bases, off = api.synth_reads(0x5EED0002, g, n_reads, 10000)
n_sym = int(off[-1]) + n_reads
t = time.time()
units = [api.build_bwt(bases, off, rev, 0) for rev in (False, True)]
idx = api.index_from_units(units[0], units[1], n_reads, n_sym); idx.upload(0)
print(f"index {n_sym/1e6:.0f} M symbols built+uploaded in {time.time()-t:.1f}s", flush=True)
p = api.params_default(5, 90); p.no_dp = no_dp
ctx = idx.ctx(p, 0)
sub_off = off[: n_correct + 1].copy(); sub = bases[: int(sub_off[-1])]
for rep in range(2):
    ctx.stats_reset()
    t = time.time()
    res, pieces = ctx.correct_reads(sub, sub_off)
    dt = time.time() - t
    st = {k: ctx.stats(v) for k, v in (("grid", K_GRID), ("seeds", K_SEEDS), ("extend", K_EXTEND), ("lf", K_LF), ("dp", K_DP), ("msa", K_MSA))}
    walks = sum(r.total_walk_num for r in res); fm = sum(r.fm_num for r in res); dpn = sum(r.dp_num for r in res)
    print(f"rep {rep}: {int(sub_off[-1])/1e6:.1f} Mbases in {dt:.2f}s = {int(sub_off[-1])/dt/1e6:.1f} Mbases/s; walks {walks} (FM {fm}, DP {dpn}); "
          + "; ".join(f"{k}: {s.launches} launches {s.total_ms:.0f} ms {s.rank_queries/1e9:.2f} G ranks" for k, s in st.items()), flush=True)
