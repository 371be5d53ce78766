"""The pile-up kernel's row-at-a-time formulation (csrc/dp_msa.hip, "wavefront-wide passes") against the step-by-step cigar walk of
MultipleAlignment::_addSequence (Thirdparty/multiple_alignment.cpp:291-377) on adversarial random cigars -- CPU only.

Both sides here are Python twins of the two paths inside dp_msa_kernel, written from the kernel's own statements: `add_row_walk`
is the walk the kernel keeps for corner-case rows (itself pinned to the oracle's restatement of the reference by the GPU tests),
`add_row_passes` the prefix-count formulation with the same fall-back conditions.  What this checks is the *derivation*: that in
the base row's old coordinates every op's effect follows from prefix counts, and that insertGapBeforeColumn's effect on the other
rows does not depend on the order of a row's insertions -- on cigars far nastier than an alignment ever produces (long I runs in
and around existing gap columns, rows starting at base 0, rows ending behind the last base).  The HIP code itself is checked
against the oracle in tests/test_gpu_fm.py::test_dp_consensus_matches_oracle and the whole-path tests."""
import random

GAP, NUL, UNSET = 4, 0xFF, 0xFE
MAX_INS = 256


class Pile:
    def __init__(self, query):
        self.T = list(query)
        self.cnt = {}                      # column -> [A, C, G, T, '-']
        for c, s in enumerate(query):
            self.col(c)[s] += 1
        self.lead, self.size = [], []
        self.lead_b = 0

    def col(self, c):
        return self.cnt.setdefault(c, [0, 0, 0, 0, 0])

    @property
    def size_b(self):
        return len(self.T)

    def state(self):
        cols = {c: tuple(v) for c, v in self.cnt.items() if any(v)}
        return (tuple(self.T), cols, tuple(self.lead), tuple(self.size), self.lead_b)

    def padded_position_of_base(self, b):
        seen = 0
        for i, t in enumerate(self.T):
            if t != GAP:
                if seen == b:
                    return i
                seen += 1
        raise AssertionError("base index beyond the row")


def add_row_walk(p, m0s, m1s, ops, S):
    """dp_msa_kernel's step-by-step walk (the `while(cig < A.n_ops)` loop)."""
    ti = p.padded_position_of_base(m0s)
    tl = p.lead_b
    il = ti + tl
    inc, cig, out = m1s, 0, []
    while cig < len(ops):
        tsym = p.T[ti] if ti < p.size_b else NUL
        op = ops[cig]
        if tsym == GAP:
            if op == 'I':
                sym = S[inc]; inc += 1; cig += 1
            else:
                sym = GAP
        elif op == 'M':
            sym = S[inc]; inc += 1; cig += 1
        elif op == 'D':
            sym = GAP; cig += 1
        else:
            c = ti + tl
            ngap = 0
            for e in range(len(p.lead)):
                le, se = p.lead[e], p.size[e]
                if c <= le:
                    p.lead[e] = le + 1
                elif c - le < se:
                    p.size[e] = se + 1; ngap += 1
            end_b = p.lead_b + p.size_b
            base_inside = False
            if c <= p.lead_b:
                p.lead_b += 1
            elif c - p.lead_b < p.size_b:
                base_inside = True
                p.T.insert(c - p.lead_b, GAP)
            if c < end_b:
                for col in range(end_b, c, -1):
                    p.cnt[col] = p.cnt.get(col - 1, [0, 0, 0, 0, 0])
            p.cnt[c] = [0, 0, 0, 0, ngap + (1 if base_inside else 0)]
            sym = S[inc]; inc += 1; cig += 1
        out.append(sym)
        ti += 1
    for n, sym in enumerate(out):
        p.col(il + n)[sym] += 1
    p.lead.append(il); p.size.append(len(out))


def add_row_passes(p, m0s, m1s, ops, S, stats):
    """dp_msa_kernel's wavefront-wide passes; returns False when the row must take the walk (nothing has been changed then)."""
    n_ops = len(ops)
    ti0 = p.padded_position_of_base(m0s)
    if n_ops == 0 or (ti0 == 0 and ops[0] == 'I'):
        return False
    tl = p.lead_b
    il = ti0 + tl
    size_b = p.size_b
    bpos, Rw = [], {}
    for i in range(ti0, size_b):
        ng = p.T[i] != GAP
        if ng:
            bpos.append(i)
        Rw[i] = UNSET if ng else GAP
    nbr = len(bpos)
    bpos.append(size_b)
    md_tot = mi_tot = 0
    run_carry = 0
    insx, inss = [], []
    last_y = None
    for j, op in enumerate(ops):
        isM, isD, isI = op == 'M', op == 'D', op == 'I'
        if not (isM or isD or isI):
            return False
        brel, inc = md_tot, m1s + mi_tot
        i_rank = j - run_carry
        if brel >= nbr:
            return False                                  # behind the last base
        pb = bpos[brel]
        fill = st = False
        if isI:
            G, y = 0, 0
            if brel > 0:
                pp = bpos[brel - 1]; G = pb - pp - 1; y = pp + 1 + i_rank
            fill = i_rank < G; st = not fill
            if st:
                y = pb - 1
        else:
            y = pb
        if st and len(insx) >= MAX_INS:
            return False
        sym = (S[inc] if inc < len(S) else 0) if (isM or isI) else GAP
        if isM or isD or fill:
            Rw[y] = sym
        if st:
            insx.append(y + 1); inss.append(sym)
        if isM or isD:
            md_tot += 1; run_carry = j + 1
        if isM or isI:
            mi_tot += 1
        last_y = y
    nin = len(insx)
    stats["ins"] += nin
    if nin:
        insg = [sum(1 for le, se in zip(p.lead, p.size) if le < x + tl and x + tl - le < se) for x in insx]
        for e in range(len(p.lead)):
            le, se = p.lead[e], p.size[e]
            p.lead[e] = le + sum(1 for x in insx if x + tl <= le)
            p.size[e] = se + sum(1 for x in insx if le < x + tl and x + tl - le < se)
        x0 = insx[0]
        T_old, cnt_old, Rw_old = list(p.T), dict(p.cnt), dict(Rw)
        p.T = p.T + [None] * nin
        for y in range(size_b - 1, x0 - 1, -1):
            sh = sum(1 for x in insx if x <= y)           # upper_bound(insx, y)
            p.cnt[y + sh + tl] = cnt_old.get(y + tl, [0, 0, 0, 0, 0])
            p.T[y + sh] = T_old[y]
            Rw[y + sh] = Rw_old[y]
        for k in range(nin):
            np_ = insx[k] + k
            p.T[np_] = GAP; Rw[np_] = inss[k]
            p.cnt[np_ + tl] = [0, 0, 0, 0, insg[k] + 1]
        assert None not in p.T
    nout = last_y + 1 - ti0 + nin
    for z in range(nout):
        sym = Rw[ti0 + z]
        assert sym <= GAP, "a position of the walked range without a symbol"
        p.col(ti0 + z + tl)[sym] += 1
    p.lead.append(il); p.size.append(nout)
    return True


def random_row(rng, p, lq):
    """(m0s, m1s, ops, S): a cigar over the base row's bases from m0s on; nastiness dialled by the mode."""
    mode = rng.random()
    m0s = 0 if mode < 0.25 else rng.randrange(lq)
    avail = lq - m0s
    n_md = rng.randint(1, avail) if mode > 0.1 else avail        # often right up to the last base
    ops = []
    if rng.random() < 0.3:
        ops += ['I'] * rng.randint(1, 4)                         # leading insertions (in front of base 0: the cursor quirk)
    for b in range(n_md):
        ops.append('M' if rng.random() < 0.7 else 'D')
        r = rng.random()
        if r < 0.35 and (b + 1 < n_md or rng.random() < 0.5):
            ops += ['I'] * (rng.randint(1, 6) if r < 0.3 else rng.randint(7, 20))
    if rng.random() < 0.05:
        ops = ['I'] * rng.randint(1, 5)
    m1s = rng.randrange(3)
    n_mi = sum(1 for o in ops if o != 'D')
    S = [rng.randrange(4) for _ in range(m1s + n_mi)]
    return m0s, m1s, ops, S


def test_row_passes_equal_the_step_walk_on_random_cigars():
    rng = random.Random(20261005)
    stats = {"ins": 0}
    rows = by_walk = 0
    for trial in range(400):
        lq = rng.randint(1, 30)
        query = [rng.randrange(4) for _ in range(lq)]
        a, b = Pile(query), Pile(query)
        for _ in range(rng.randint(1, 12)):
            m0s, m1s, ops, S = random_row(rng, a, lq)
            add_row_walk(a, m0s, m1s, ops, S)
            before = b.state()
            if not add_row_passes(b, m0s, m1s, ops, S, stats):
                assert b.state() == before                 # a refused row has touched nothing
                add_row_walk(b, m0s, m1s, ops, S)
                by_walk += 1
            rows += 1
            assert a.state() == b.state(), (trial, m0s, m1s, "".join(ops))
    # the generator reaches both paths and plenty of structural insertions and gap fills
    assert rows > 2000 and 0.05 < by_walk / rows < 0.6 and stats["ins"] > 5000
