"""Debug: per-sweep trace of one read from the CPU harness (mode host -> tools/_host_trace.bin) and from the GPU kernel
(mode gpu: runs the kernel with LRSC_SM_TRACE and prints the first record that differs from the host trace)."""
import ctypes as C, os, sys, tempfile
from pathlib import Path
import numpy as np
REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))
from longreadselfcorrect_amd import Lrsc
from oracle import oracle_py
from tests.conftest import Dataset

mode, n, rd = sys.argv[1], 70, int(sys.argv[2]) if len(sys.argv) > 2 else 0
api, orc = Lrsc(), oracle_py.Oracle()
HOST = REPO / "tools" / "_host_trace.bin"
with tempfile.TemporaryDirectory() as tmp:
    ds = Dataset(api, orc, tmp, 4000, 180, 2000)
    off = ds.off[: n + 1].copy(); bases = ds.bases[: int(off[-1])]
    p = api.params_default(5, 90); p.no_dp = 1
    if mode == "host":
        from tests.emul import Emul
        em = Emul()
        u = [np.fromfile(f"{ds.prefix}.{ext}", dtype=np.uint8)[30:] for ext in ("bwt", "rbwt")]
        h = em.index(u[0], u[1], int(ds.off[-1]) + ds.n_reads, tables=(5, 9, 11))
        ob, orb = orc.bwt_load(ds.prefix + ".bwt"), orc.bwt_load(ds.prefix + ".rbwt")
        count, seeds, _ = orc.find_seeds(ob, orb, p, bases, off)
        buf = np.zeros(14 * 400000 + 1, dtype=np.uint32)
        em.lib.emul_set_trace(buf.ctypes.data_as(C.c_void_p), buf.size, rd)
        em.correct_reads(h, p, bases, off, count, seeds)
        em.lib.emul_trace_words.restype = C.c_uint32
        w = em.lib.emul_trace_words()
        buf[0] = w
        buf[:w].tofile(HOST)
        print("host trace:", (w - 1) // 14, "sweeps")
    else:
        idx = api.index_open(ds.prefix + ".bwt", ds.prefix + ".rbwt"); idx.upload(0)
        os.environ["LRSC_SM_TRACE"] = "/tmp/gpu_trace.bin"; os.environ["LRSC_SM_TRACE_READ"] = str(rd)
        ctx = idx.ctx(p, 0)
        ctx.correct_reads(bases, off)
        ctx.close()
        g = np.fromfile("/tmp/gpu_trace.bin", dtype=np.uint32); hst = np.fromfile(HOST, dtype=np.uint32)
        G = g[1:g[0]].reshape(-1, 14); H = hst[1:hst[0]].reshape(-1, 14)
        # the GPU lane also records sweeps spent waiting for the set-up quorum (pc == PC_NEXT without progress): drop repeats
        def dedup(T):
            keep = [0] + [i for i in range(1, len(T)) if not (T[i, 1] == 0 and (T[i] == T[i - 1]).all())]
            return T[keep]
        def norm(T):
            T = T.copy()
            none, tab = T[:, 1] == 0, T[:, 1] == 2
            T[none, 2:] = 0
            T[tab, 2:4] = 0; T[tab, 6:10] = 0
            rank = T[:, 1] == 1
            T[rank, 4:6] = 0
            return T
        G, H = dedup(norm(G)), dedup(norm(H))
        print("gpu sweeps", len(G), "host sweeps", len(H))
        m = min(len(G), len(H))
        d = np.flatnonzero((G[:m] != H[:m]).any(axis=1))
        print("first differing sweep:", d[:1])
        if len(d):
            i = int(d[0])
            for j in range(max(0, i - 3), min(m, i + 3)):
                print(j, "G", G[j].tolist()); print(j, "H", H[j].tolist())
