"""CPU checks of the oracle's FM-extend / DP layers: pinned to the reference's object code where it
compiles (IntervalTree, Overlapper::extendMatch), semantic checks elsewhere."""
import numpy as np
import pytest


def _mutate(rng, s, p_del=0.045, p_sub=0.015, p_ins=0.09):
    out = []
    for c in s:
        u = rng.random()
        if u >= p_del:
            out.append(rng.choice([x for x in "ACGT" if x != c]) if u < p_del + p_sub else c)
        while rng.random() < p_ins / (1 + p_ins):
            out.append(rng.choice(list("ACGT")))
    return "".join(out)


@pytest.mark.parametrize("n", [1, 7, 8, 16, 17, 40, 150, 300, 600])
def test_interval_tree_matches_reference_including_sort_tie_order(ref, oracle, n):
    """Entries with equal start come back in the order libstdc++'s introsort leaves them
    (IntervalTree.cpp:18) -- the order is observable in isSupportedByNewSeed, so it must match exactly."""
    rng = np.random.default_rng(n)
    for trial in range(6):
        # few distinct k-mer intervals, many query offsets each: exactly the shape of the 9-mer / 5-mer trees
        n_keys = max(1, n // rng.integers(1, 12))
        lo = np.sort(rng.integers(1000, 10**7, size=n_keys))
        width = rng.integers(0, 50, size=n_keys)
        key = rng.integers(0, n_keys, size=n)
        start, stop, value = lo[key], lo[key] + width[key], np.arange(n)
        queries = [(lo[k], lo[k] + width[k]) for k in range(n_keys)]
        queries += [(lo[k] + 1, lo[k] + max(1, width[k]) - 1) for k in range(n_keys) if width[k] > 2]   # strict sub-interval
        queries += [(5, 6), (10**8, 10**8 + 1)]
        got = oracle.itree_query_all(start, stop, value, queries)
        want = ref.itree_query_all(start, stop, value, queries)
        for g, w in zip(got, want):
            np.testing.assert_array_equal(g, w)


def test_extend_match_matches_reference(ref, oracle):
    """Banded DP of the fallback (Overlapper::extendMatch, +1/-1/-8, band 200) incl. the homopolymer
    tie-break rules and reads at s[size()]."""
    rng = np.random.default_rng(42)
    cases = []
    for L in (30, 120, 300, 700):
        truth = "".join(rng.choice(list("ACGT"), size=L))
        for _ in range(6):
            a, b = _mutate(rng, truth), _mutate(rng, truth)
            cases.append((a, b, 0, 0))
            k = 15
            a2, b2 = a + truth[-k:], b + truth[-k:]
            cases.append((a2, b2, len(a2) - k, len(b2) - k))          # the isRC flavour: anchored at the tail
    cases += [("A" * 50 + "C" * 20, "A" * 44 + "C" * 26, 0, 0), ("ACGT" * 20, "ACGT" * 18 + "AC", 0, 0),
              ("AAAAAAAAAA", "AAAAAAA", 0, 0), ("ACGTACGTAC", "ACGTTACGTAC", 0, 0)]
    for s1, s2, a, b in cases:
        assert oracle.extend_match(s1, s2, a, b) == ref.extend_match(s1, s2, a, b)


@pytest.fixture(scope="module")
def orc_index(oracle, small_ds):
    ob, orb = oracle.bwt_load(small_ds.prefix + ".bwt"), oracle.bwt_load(small_ds.prefix + ".rbwt")
    yield ob, orb
    ob.close(); orb.close()


def _kmer_accuracy(seq: str, genome_kmers: set, k: int = 21) -> float:
    n = len(seq) - k + 1
    if n <= 0:
        return 0.0
    return sum(seq[i:i + k] in genome_kmers for i in range(n)) / n


def _genome_kmers(genome: np.ndarray, k: int = 21) -> set:
    g = genome.tobytes().decode()
    comp = str.maketrans("ACGT", "TGCA")
    rc = g.translate(comp)[::-1]
    return {s[i:i + k] for s in (g, rc) for i in range(len(s) - k + 1)}


@pytest.mark.parametrize("no_dp", [1, 0])
def test_whole_path_corrects_reads(api, oracle, small_ds, orc_index, no_dp):
    """Semantic check of the restated pipeline: 15%-error reads come out far closer to the genome."""
    ob, orb = orc_index
    p = api.params_default(5, 90)
    p.no_dp = no_dp
    n = 12
    off = small_ds.off[: n + 1].copy()
    bases = small_ds.bases[: int(off[-1])]
    run = oracle.correct_reads(ob, orb, p, bases, off)
    gk = _genome_kmers(small_ds.genome)
    raw = small_ds.reads[:n]
    recs = run.correct_fa.strip().split("\n")
    ids, seqs = recs[0::2], recs[1::2]
    assert len(ids) >= n - 2 and all(i.startswith(">r") for i in ids)          # almost every read gets >= 2 seeds
    raw_acc = np.mean([_kmer_accuracy(r, gk) for r in raw])
    cor_acc = np.mean([_kmer_accuracy(s, gk) for s in seqs])
    assert raw_acc < 0.15 and cor_acc > 0.80, (raw_acc, cor_acc)
    c = run.counters
    assert (c[:, 7] + c[:, 8] <= c[:, 3]).all()                                # FMNum + DPNum <= totalWalkNum
    assert c[:, 7].sum() > 0.5 * c[:, 3].sum()                                 # most walks solved by FM-extend
    if no_dp:
        assert c[:, 8].sum() == 0
    w = run.walks
    assert set(np.unique(w[:, 3])) <= {1, -1, -2, -3}
    # stats block carries the integer lines of the reference's stdout block
    assert run.stats.startswith("TotalReadsLen: ") and "\nFMNum: " in run.stats
    # deterministic
    run2 = oracle.correct_reads(ob, orb, p, bases, off)
    assert run2.correct_fa == run.correct_fa and run2.discard_fa == run.discard_fa
    run.close(); run2.close()


def test_single_walk_bridges_a_gap(api, oracle, small_ds, orc_index):
    """One LongReadSelfCorrectByOverlap walk between two true genome k-mers across a noisy read segment."""
    ob, orb = orc_index
    p = api.params_default(5, 90)
    g = small_ds.genome.tobytes().decode()
    rng = np.random.default_rng(9)
    ok = 0
    for t in range(8):
        s = int(rng.integers(100, len(g) - 600))
        src, gap, trg = g[s:s + 19], g[s + 19:s + 19 + 200], g[s + 219:s + 219 + 19]
        noisy = _mutate(rng, gap)
        code, merged, st = oracle.extend_walk(ob, orb, p, src, noisy, trg, len(noisy), 17, 19, 3)
        assert code in (1, -1, -2, -3)
        if code == 1:
            ok += 1
            assert merged.startswith(src[-17:]) and merged.endswith(trg)
            # the walk reproduces the true genome segment (or a near-identical path)
            truth = src[-17:] + gap + trg
            assert abs(len(merged) - len(truth)) <= 6
            assert st[0] > 150
    assert ok >= 5
