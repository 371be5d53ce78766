#!/usr/bin/env python3
"""configs[3]/[4] readiness on one GPU: a read set ABOVE 2^32 symbols per strand.

    python3 tools/big_index_check.py [--reads 420000] [--sample 12]

1. synthetic reads (10 kb templates, 15 % error, 90x over a random genome), > 2^32 symbols;
2. both BWTs by the GPU builder (grouped jobs, 64-bit positions) -- timed;
3. the index goes to HBM in the Block64 layout (128 symbols per 64-byte block, 64-bit counters) with its 5/9/13/15-mer tables;
4. size-independent check of BWT + rank structures: LF-walking from row r (the r-th sentinel) must spell read r backwards --
   on a sample of rows, 10 k dependent rank steps each, through the device's own Block64 path;
5. seeds and whole-path correction of a sample of reads on the GPU against the CPU oracle over the same BWT (bit-exact);
6. seed-stage / correction timing of one 20k-read batch through the wide path, and the bytes per symbol actually resident.
Writes gpurun_out/big_index.json and progress lines to gpurun_out/big_index.log.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
OUT = ROOT / "gpurun_out"
OUT.mkdir(exist_ok=True)
T0 = time.time()


def log(msg):
    line = f"[{time.time() - T0:7.1f}s] {msg}"
    print(line, flush=True)
    with open(OUT / "big_index.log", "a") as f:
        f.write(line + "\n")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=420_000)
    ap.add_argument("--read-len", type=int, default=10_000)
    ap.add_argument("--sample", type=int, default=12)
    ap.add_argument("--batch", type=int, default=20_000)
    a = ap.parse_args()
    os.environ.setdefault("LRSC_BWT_PROFILE", "1")
    import __graft_entry__ as g

    g.build_product()
    from longreadselfcorrect_amd import Lrsc
    from oracle import oracle_py

    oracle_py.build_oracle()
    api, orc = Lrsc(), oracle_py.Oracle()
    res = {"reads": a.reads, "read_len": a.read_len}
    genome_len = int(a.reads * a.read_len * 1.045 / 90)
    log(f"genome {genome_len / 1e6:.1f} Mb, {a.reads} reads")
    genome = api.synth_genome(0xB16, genome_len)
    bases, off = api.synth_reads(0xB17, genome, a.reads, a.read_len)
    n_sym = int(off[-1]) + a.reads
    res["num_symbols"] = n_sym
    res["above_2_32"] = n_sym >= 2 ** 32
    log(f"{n_sym} symbols per strand ({n_sym / 2**32:.3f} x 2^32)")
    units = []
    for rev in (False, True):
        t = time.time()
        u = api.build_bwt(bases, off, rev, 0)
        dt = time.time() - t
        units.append(u)
        res["build_rbwt_s" if rev else "build_bwt_s"] = round(dt, 2)
        log(f"{'rbwt' if rev else 'bwt'}: {u.size} RL units in {dt:.1f} s ({n_sym / dt / 1e6:.0f} M symbols/s incl. transfers and run-length coding)")
    t = time.time()
    index = api.index_from_units(units[0], units[1], a.reads, n_sym)
    index.upload(0)
    info = index.info()
    res.update(block_symbols=int(info.block_symbols), block_bytes=int(info.block_bytes), device_bytes=int(info.device_bytes),
               runs=[int(info.num_runs[0]), int(info.num_runs[1])])
    log(f"index in HBM: {info.block_symbols} symbols per {info.block_bytes}-byte block, {info.device_bytes / 1e9:.2f} GB resident "
        f"({info.device_bytes / n_sym:.3f} B/symbol for both strands + tables), {time.time() - t:.1f} s")
    assert info.block_symbols == 128, "expected the Block64 layout above 2^32 symbols"
    p = api.params_default(10, 90)
    ctx = index.ctx(p, 0)

    # 4. LF-walk: row r spells read r backwards
    rng = np.random.default_rng(1)
    rows = np.unique(np.concatenate([[0, 1, a.reads - 1], rng.integers(0, a.reads, size=29)])).astype(np.uint64)
    lens = (off[rows.astype(np.int64) + 1] - off[rows.astype(np.int64)]).astype(np.uint32)
    got = ctx.lf_walk(rows, np.zeros(rows.size, dtype=np.uint8), lens + 5)
    ok = 0
    for r, s in zip(rows.tolist(), got):
        want = bases[int(off[r]): int(off[r + 1])].tobytes().decode()[::-1]
        ok += s == want
    res["lf_walk_reads_checked"] = int(rows.size)
    res["lf_walk_reads_identical"] = int(ok)
    log(f"LF-walk through the Block64 index: {ok} of {rows.size} reads spelled back exactly ({int(lens.sum())} dependent rank steps)")
    assert ok == rows.size
    # the reversed-read index: row r spells read r forwards
    got = ctx.lf_walk(rows[:8], np.ones(8, dtype=np.uint8), lens[:8] + 5)
    ok_r = sum(s == bases[int(off[r]): int(off[r + 1])].tobytes().decode() for r, s in zip(rows[:8].tolist(), got))
    res["lf_walk_rbwt_identical"] = int(ok_r)
    assert ok_r == 8

    # 5. sampled whole path vs the oracle
    sample = np.unique(np.concatenate([[0, 1], rng.integers(0, a.reads, size=a.sample - 2)]))
    sb = np.concatenate([bases[int(off[r]): int(off[r + 1])] for r in sample])
    so = np.zeros(sample.size + 1, dtype=np.uint64)
    so[1:] = np.cumsum([int(off[r + 1] - off[r]) for r in sample])
    t = time.time()
    ob = orc.bwt_from_units(units[0], a.reads, n_sym)
    orb = orc.bwt_from_units(units[1], a.reads, n_sym)
    log(f"oracle BWTs loaded in {time.time() - t:.1f} s")
    b = ctx.batch(sb, so)
    b.find_seeds()
    count, seeds, attr = b.seeds()
    wcount, wseeds, wattr = orc.find_seeds(ob, orb, p, sb, so)
    gs = np.stack([seeds[f] for f in seeds.dtype.names], axis=1).astype(np.int32)
    seeds_ok = bool(np.array_equal(count, wcount) and np.array_equal(gs, wseeds) and np.array_equal(attr, wattr))
    log(f"seeds of {sample.size} sampled reads: {int(count.sum())} seeds, identical to the oracle: {seeds_ok}")
    results, poff, outb = b.correct()
    b.close()
    t = time.time()
    want = orc.correct_reads(ob, orb, p, sb, so)
    log(f"oracle corrected the sample in {time.time() - t:.1f} s")
    names = ("total_reads_len", "corrected_len", "total_seed_num", "total_walk_num", "high_error_num", "exceed_depth_num",
             "exceed_leave_num", "fm_num", "dp_num", "seed_dis", "merge")
    gc = np.array([[getattr(r, f) for f in names] for r in results], dtype=np.int64)
    buf = outb.tobytes()
    cfa = "".join(f">r{i}\n{buf[int(poff[r.piece_first]): int(poff[r.piece_first + 1])].decode()}\n" for i, r in enumerate(results) if r.merge)
    correct_ok = bool(np.array_equal(gc, want.counters) and cfa == want.correct_fa)
    res.update(sample_reads=int(sample.size), seeds_identical=seeds_ok, correct_identical=correct_ok,
               sample_counters=gc.sum(axis=0).tolist())
    log(f"whole path of the sample: counters + correct.fa identical to the oracle: {correct_ok}  (sums {gc.sum(axis=0).tolist()})")
    want.close(); ob.close(); orb.close()
    assert seeds_ok and correct_ok

    # 6. one batch through the wide path
    nb = min(a.batch, a.reads)
    bo = off[: nb + 1].copy()
    bb = bases[: int(bo[-1])]
    b = ctx.batch(bb, bo)
    ctx.stats_reset()
    t = time.time(); b.find_seeds(); ctx.sync(); t_seed = time.time() - t
    t = time.time(); b.correct(); t_corr = time.time() - t
    b.close()
    mb = int(bo[-1]) / 1e6
    res.update(batch_reads=nb, seed_stage_mbases_s=round(mb / t_seed, 1), correct_mbases_s=round(mb / (t_seed + t_corr), 2))
    log(f"{nb}-read batch through Block64: seed stage {mb / t_seed:.0f} Mbases/s, seeds+correction {mb / (t_seed + t_corr):.1f} corrected Mbases/s")
    ctx.close(); index.close()
    (OUT / "big_index.json").write_text(json.dumps(res, indent=1))
    log("done")


if __name__ == "__main__":
    main()
