"""SURVEY section 8 row f4 on the GPU (`-m gpu`): what `--debugseed` / `--onlyseed -b` write and the `kmerfreq` / `kmercheck`
tools print, against the CPU oracle on the same reads (the barcode verdicts against the host BCode that
tests/test_host_tools.py pins to the reference's object code)."""
from __future__ import annotations

import subprocess

import numpy as np
import pytest

from .conftest import REPO, write_fasta

pytestmark = pytest.mark.gpu
STRIDE = REPO / "longreadselfcorrect_amd" / "_build" / "stride"


def _expected_walk_log(count, seeds, walks):
    """One byte per seed from the oracle's walk records (read, srcStart, trgStart, code, via)."""
    first = np.concatenate([[0], np.cumsum(count)]).astype(np.int64)
    log = np.zeros(int(first[-1]), dtype=np.uint8)
    for r, src, trg, code, via in walks.tolist():
        if code > 0:
            continue
        s = seeds[first[r]: first[r + 1]]
        t = int(np.flatnonzero(s[:, 0] == trg)[0])
        assert t >= 1 and s[t - 1, 0] == src                       # the walk's source is the seed before its target
        log[first[r] + t] = (code + 4) | (16 if via == 2 else 0)
    return log


@pytest.mark.parametrize("kernel", ["one-kernel", "two-class"])
@pytest.mark.parametrize("ds_name,nodp,split", [("repeat_ds", 0, 0), ("repeat_ds", 1, 1), ("small_ds", 0, 1)])
def test_debug_collection_matches_oracle(api, oracle, request, ds_name, nodp, split, kernel, monkeypatch):
    """lrsc_batch_set_debug: dropped (hitchhiking) seeds, the repeat ratio per position and the failed-walk log."""
    if kernel == "two-class":
        monkeypatch.setenv("LRSC_WP_SCHED", "1")
    ds = request.getfixturevalue(ds_name)
    n = 200
    off = ds.off[: n + 1].copy()
    bases = ds.bases[: int(off[-1])]
    p = api.params_default(5, 90)
    p.no_dp, p.split = nodp, split
    ob, orb = oracle.bwt_load(ds.prefix + ".bwt"), oracle.bwt_load(ds.prefix + ".rbwt")
    idx = api.index_open(ds.prefix + ".bwt", ds.prefix + ".rbwt")
    idx.upload(0)
    ctx = idx.ctx(p, 0)
    b = ctx.batch(bases, off)
    b.set_debug(b.DEBUG_OUTCASTS | b.DEBUG_WALKS | b.DEBUG_RATIO)
    b.find_seeds()
    count, seeds, _ = b.seeds(want_attribute=False)
    ocount, outcasts = b.outcast_seeds()
    ratio = b.repeat_ratio()
    res, _, _ = b.correct()
    log = b.walk_log(int(count.sum()))
    b.close(); ctx.close(); idx.close()

    wcount, wseeds, _ = oracle.find_seeds(ob, orb, p, bases, off)
    wocount, woutcasts, wratio = oracle.find_seeds_debug(ob, orb, p, bases, off)
    np.testing.assert_array_equal(count, wcount)
    np.testing.assert_array_equal(ocount, wocount)
    got = np.stack([outcasts[f] for f in outcasts.dtype.names], axis=1).astype(np.int32)
    np.testing.assert_array_equal(got[:, :4], woutcasts[:, :4])     # the four fields the .seed file prints (+ position / length)
    np.testing.assert_array_equal(ratio.view(np.uint32), wratio.view(np.uint32))      # bit-exact floats
    run = oracle.correct_reads(ob, orb, p, bases, off)
    want_log = _expected_walk_log(wcount, wseeds, run.walks)
    np.testing.assert_array_equal(log, want_log)
    assert all(r.status == 0 for r in res)
    if ds_name == "repeat_ds":
        assert int(wocount.sum()) > 0 and (want_log > 0).sum() > 0
        if nodp:
            assert ((want_log & 16) > 0).sum() == (want_log > 0).sum()      # --nodp: every failed walk is a failed "DP" too
    run.close(); ob.close(); orb.close()


def _write_barcodes(path, ds, n, rng):
    """Synthetic blocks (two per read, the second on the reverse strand) in the nine-column layout."""
    from tests.golden.make_host_tools import make_block

    lines, blocks = [], {}
    reads = ds.reads[:n]
    for i, seq in enumerate(reads):
        L = len(seq)
        if L < 400:
            continue
        cut = L // 2
        for (s, e), rvc in (((10, cut - 5), False), ((cut + 5, L - 10), True)):
            code = make_block(rng, seq_len=e - s + 60)["code"][: 2 * (e - s)]
            lines.append(f"r{i} {s} {e} ref {s} {e} {code} {'True' if rvc else 'False'} 0")
            blocks.setdefault(f"r{i}", []).append(dict(start=s, end=e, code=code, rvc=int(rvc)))
    path.write_text("\n".join(lines) + "\n")
    return blocks


@pytest.fixture(scope="module")
def driver(tmp_path_factory):
    exe = tmp_path_factory.mktemp("host_tools_gpu") / "driver"
    subprocess.run(["g++", "-std=c++14", "-O2", "-o", str(exe), str(REPO / "tests/host_tools/driver.cpp"),
                    str(REPO / "longreadselfcorrect_amd/host/BCode.cpp"), "-lz"], check=True)
    return str(exe)


def _freqs(ob, orb, words):
    """BiBWTInterval::getFreq of equally long words: reverse(w) in the .rbwt index + reverse-complement(w) in the .bwt index."""
    k = len(words[0])
    comp = str.maketrans("ACGT", "TGCA")
    f = orb.find_intervals(np.frombuffer("".join(w[::-1] for w in words).encode(), dtype=np.uint8), k)
    v = ob.find_intervals(np.frombuffer("".join(w[::-1].translate(comp) for w in words).encode(), dtype=np.uint8), k)
    return np.maximum(0, f[:, 1] - f[:, 0] + 1) + np.maximum(0, v[:, 1] - v[:, 0] + 1)


def _seed_lines(seq, seeds):
    return "".join(f"{seq[s[0]: s[0] + s[1]]}\t{s[2]}\t{s[0]}\t{'Yes' if s[3] else 'No'}\n" for s in seeds.tolist())


def test_pbcorrect_debugseed_writes_the_reference_files(api, oracle, repeat_ds, tmp_path):
    ds, n = repeat_ds, 150
    reads = ds.reads[:n] + ["ACGTACGTAC"]                       # + one read shorter than the start k-mer: no files for it
    fa = tmp_path / "reads.fa"
    write_fasta(fa, reads)
    out = tmp_path / "out"
    r = subprocess.run([str(STRIDE), "pbcorrect", "-p", ds.prefix, "-o", str(out), "-c", "90", "-g", "5", "--batch", "64", "--debugseed",
                        "--debugextend", str(fa)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    from oracle.oracle_py import pack_reads

    bases, off = pack_reads(reads)
    p = api.params_default(5, 90)
    ob, orb = oracle.bwt_load(ds.prefix + ".bwt"), oracle.bwt_load(ds.prefix + ".rbwt")
    count, seeds, _ = oracle.find_seeds(ob, orb, p, bases, off)
    ocount, outcasts, ratio = oracle.find_seeds_debug(ob, orb, p, bases, off)
    run = oracle.correct_reads(ob, orb, p, bases, off)
    assert (out / "correct.fa").read_text() == run.correct_fa and (out / "discard.fa").read_text() == run.discard_fa
    first = np.concatenate([[0], np.cumsum(count)]).astype(np.int64)
    ofirst = np.concatenate([[0], np.cumsum(ocount)]).astype(np.int64)
    walks = run.walks
    n_ext = n_dp = 0
    for i, seq in enumerate(reads):
        rid = f"r{i}"
        if len(seq) < p.start_kmer_len:
            assert not (out / "seed" / f"{rid}.seed").exists() and not (out / "extend" / f"{rid}.log").exists()
            continue
        assert (out / "seed" / f"{rid}.seed").read_text() == _seed_lines(seq, seeds[first[i]: first[i + 1]])
        err = out / "seed" / "error" / f"{rid}.seed"
        if count[i] + ocount[i] >= 2:
            assert err.read_text() == _seed_lines(seq, outcasts[ofirst[i]: ofirst[i + 1]])
        else:
            assert not err.exists()
        # extend/<id>.log: "pos<TAB>ratio" with the stream's default float formatting (6 significant digits)
        got = (out / "extend" / f"{rid}.log").read_text().split("\n")[:-1]
        want = [f"{pos}\t{float(x):g}" for pos, x in enumerate(ratio[int(off[i]): int(off[i + 1])])]
        assert got == want
        ext, dp = out / "extend" / f"{rid}.ext", out / "extend" / f"{rid}.dp"
        if count[i] < 2:
            assert not ext.exists() and not dp.exists()
            continue
        w = walks[(walks[:, 0] == i) & (walks[:, 3] < 0)]
        assert ext.read_text() == "".join(f"{a}\t{b}\t{c + 4}\n" for _, a, b, c, _ in w.tolist())
        assert dp.read_text() == "".join(f"{a}\t{b}\n" for _, a, b, c, via in w.tolist() if via == 2)
        n_ext += len(w); n_dp += int((w[:, 4] == 2).sum())
    assert n_ext > 20 and int(ocount.sum()) > 0
    run.close(); ob.close(); orb.close()


def test_pbcorrect_onlyseed_validates_seeds_against_the_barcode(api, oracle, repeat_ds, driver, tmp_path):
    ds, n = repeat_ds, 120
    fa = tmp_path / "reads.fa"
    write_fasta(fa, ds.reads[:n])
    blocks = _write_barcodes(tmp_path / "barcode.txt", ds, n, np.random.default_rng(77))
    out = tmp_path / "out"
    r = subprocess.run([str(STRIDE), "pbcorrect", "-p", ds.prefix, "-o", str(out), "-c", "90", "-g", "5", "--batch", "50", "--onlyseed", "-b",
                        str(tmp_path / "barcode.txt"), str(fa)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert not (out / "correct.fa").exists() and (out / "seed" / "r0.seed").exists() and not (out / "extend" / "r0.ext").exists()
    p = api.params_default(5, 90)
    ob, orb = oracle.bwt_load(ds.prefix + ".bwt"), oracle.bwt_load(ds.prefix + ".rbwt")
    off = ds.off[: n + 1].copy()
    count, seeds, _ = oracle.find_seeds(ob, orb, p, ds.bases[: int(off[-1])], off)
    first = np.concatenate([[0], np.cumsum(count)]).astype(np.int64)
    reads = ds.reads[:n]
    # verdict of every seed that lies inside a block, by the (reference-pinned) host BCode through the test driver
    ask, where = [], []
    status = np.zeros((n, 3), dtype=np.int64)
    for i in range(n):
        for s in seeds[first[i]: first[i + 1]].tolist():
            blk = next((b for b in blocks.get(f"r{i}", []) if s[0] >= b["start"] and s[0] + s[1] - 1 <= b["end"]), None)
            if blk is None:
                status[i, 2] += 1
            else:
                ask.append(f"{s[0]} {s[1]} {blk['start']} {blk['end']} {blk['rvc']} {blk['code']} {reads[i]}\n")
                where.append(i)
    verdicts = subprocess.run([driver, "validate"], input="".join(ask), capture_output=True, text=True, check=True).stdout.split()
    for i, v in zip(where, verdicts):
        assert v in ("0", "1")
        status[i, 0 if v == "1" else 1] += 1

    def line(name, st):
        tot = int(st.sum())
        return "%s [%d] %.2f%% %.2f%% %.2f%%\n" % (name, tot, 100 * st[0] / tot, 100 * st[1] / tot, 100 * st[2] / tot) if st[1] > 0 else ""

    assert (out / "total.seed").read_text() == "".join(line(f"r{i}", status[i]) for i in range(n))
    assert r.stdout == line("TOTAL", status.sum(axis=0))
    assert status[:, 1].sum() > 0 and status[:, 0].sum() > 0
    ob.close(); orb.close()


def test_kmerfreq_tool(api, oracle, small_ds):
    ds = small_ds
    q1, q2 = ds.reads[0][100:160], ds.reads[3][20:75]
    r = subprocess.run([str(STRIDE), "kmerfreq", "-p", ds.prefix, "-c", "90"], input=f"{q1} 19 1\n{q2} 15 2\n", capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "Please enter query sequence, kmer size and mode:" in r.stderr and "Exit successfully!" in r.stderr
    ob, orb = oracle.bwt_load(ds.prefix + ".bwt"), oracle.bwt_load(ds.prefix + ".rbwt")
    thr = oracle.threshold_table_range(90, 100)

    def freq(w):
        return int(_freqs(ob, orb, [w])[0])

    want = ""
    for q, k, mode in ((q1, 19, 1), (q2, 15, 2)):
        for pos in range(len(q) - k + 1):
            want += f"{pos}\t{q[pos: pos + k]}\t{freq(q[pos: pos + k])} <-> {float(thr[mode, k]):g}\t{q[: k + pos]}\t{freq(q[: k + pos])} <-> {float(thr[mode, k + pos]):g}\n"
        want += "-\n"
    assert r.stdout == want
    ob.close(); orb.close()


def test_kmercheck_tool(api, oracle, small_ds, driver, tmp_path):
    ds, n = small_ds, 40
    fa = tmp_path / "reads.fa"
    write_fasta(fa, ds.reads[:n])
    blocks = _write_barcodes(tmp_path / "barcode.txt", ds, n, np.random.default_rng(5))
    out = tmp_path / "box"
    r = subprocess.run([str(STRIDE), "kmercheck", "-p", ds.prefix, "-o", str(out), "-b", str(tmp_path / "barcode.txt"), "-c", "90", "-l", "15", "-u",
                        "23", "-s", "4", "--batch", "16", str(fa)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "Using kmer size : 15 - 23 (4)" in r.stderr
    ob, orb = oracle.bwt_load(ds.prefix + ".bwt"), oracle.bwt_load(ds.prefix + ".rbwt")
    reads = ds.reads[:n]
    total, value = "", ""
    for k in (15, 19, 23):
        words, ask = [], []
        for i in range(n):
            for b in blocks.get(f"r{i}", []):
                for pos in range(b["start"], b["end"] - k + 1):
                    words.append(reads[i][pos: pos + k])
                    ask.append(f"{pos} {k} {b['start']} {b['end']} {b['rvc']} {b['code']} {reads[i]}\n")
        freq = _freqs(ob, orb, words)
        verdicts = subprocess.run([driver, "validate"], input="".join(ask), capture_output=True, text=True, check=True).stdout.split()
        crt = [int(x) for x, ok in zip(freq, verdicts) if x != 1 and ok == "1"]
        err = [int(x) for x, ok in zip(freq, verdicts) if x != 1 and ok == "0"]
        assert crt and err and freq.min() >= 1
        txt = subprocess.run([driver, "compare", "90", str(k)], input=" ".join(map(str, crt)) + "\n" + " ".join(map(str, err)) + "\n",
                             capture_output=True, text=True, check=True).stdout.split("\n")
        total += txt[0] + "\n"; value += txt[1] + "\n"
    assert (out / "total.box").read_text() == total and (out / "value.box").read_text() == value
    ob.close(); orb.close()
