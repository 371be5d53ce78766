// GlobalAlign.h -- the banded global alignment SAIPBSelfCorrectTree uses to choose among several merged sequences: stdaln's
// aln_stdaln(s1, s2, &aln_param_pacbio, ALN_TYPE_GLOBAL, 1) of the reference (Thirdparty/stdaln.c:231-248,364-546,780-862), reduced
// to what its caller reads: the number of aligned equal bases (PacBio/SAIPBSelfCTree.cpp:186-194).
// Affine gaps (open 1, extend 1, free end gaps), match +1, mismatch -8, N -2, band = 50 + the length difference.  The three
// score layers are kept per row, the predecessor choice of every cell in one byte; ties are broken as in stdaln (M over I over D
// for a match state, "open" only when strictly better).  Checked against the reference's object code through tests/golden/stdaln_kats.json.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace stride {

struct GlobalAlignment { int matches, score, columns; };

inline GlobalAlignment globalAlignPacBio(const std::string& a, const std::string& b)
{
    GlobalAlignment out{0, 0, 0};
    const int n1 = (int)a.size(), n2 = (int)b.size();
    if(n1 == 0 || n2 == 0) return out;
    const int kInf = -1073741823, open = 1, ext = 1, endExt = 0, band = 50;
    auto code = [](char c) -> int { switch(c) { case 'A': case 'a': return 0; case 'G': case 'g': return 1; case 'C': case 'c': return 2; case 'T': case 't': return 3; default: return 4; } };
    std::vector<uint8_t> x(n1 + 1), y(n2 + 1);
    for(int i = 1; i <= n1; ++i) x[i] = (uint8_t)code(a[i - 1]);
    for(int j = 1; j <= n2; ++j) y[j] = (uint8_t)code(b[j - 1]);
    auto sub = [](int p, int q) { return (p == 4 || q == 4) ? -2 : (p == q ? 1 : -8); };
    int w1 = n1 > n2 ? n1 - n2 + band : band, w2 = n1 > n2 ? band : n2 - n1 + band;
    if(w1 > n1) w1 = n1;
    if(w2 > n2) w2 = n2;
    // back pointers: bits 0-1 where M came from, bits 2-3 where I came from, bits 4-5 where D came from (0 = M, 1 = I, 2 = D)
    std::vector<uint8_t> back((size_t)(n2 + 1) * (n1 + 1), 0);
    auto at = [&](int j, int i) -> uint8_t& { return back[(size_t)j * (n1 + 1) + i]; };
    struct Cell { int m, i, d; };
    std::vector<Cell> r0(n1 + 1, Cell{0, 0, 0}), r1(n1 + 1, Cell{0, 0, 0});
    Cell* cur = r0.data();
    Cell* prev = r1.data();
    auto fromDiag = [&](Cell& c, uint8_t& bp, const Cell& p, int s) {
        int from;
        if(p.m >= p.i) from = p.m >= p.d ? 0 : 2; else from = p.i > p.d ? 1 : 2;
        c.m = (from == 0 ? p.m : from == 1 ? p.i : p.d) + s;
        bp = (uint8_t)((bp & ~3u) | (unsigned)from);
    };
    auto fromAbove = [&](Cell& c, uint8_t& bp, const Cell& p, int e) {          // the I layer: a gap in the first sequence
        const bool opens = p.m - open > p.i;
        c.i = opens ? p.m - open - e : p.i - e;
        bp = (uint8_t)((bp & ~12u) | (opens ? 0u : 4u));
    };
    auto fromLeft = [&](Cell& c, uint8_t& bp, const Cell& p, int e) {           // the D layer: a gap in the second sequence
        const bool opens = p.m - open > p.d;
        c.d = opens ? p.m - open - e : p.d - e;
        bp = (uint8_t)((bp & ~48u) | (opens ? 0u : 32u));
    };
    auto blank = [&](Cell& c) { c.m = c.i = c.d = kInf; };
    auto flip = [&]() { Cell* t = cur; cur = prev; prev = t; };

    blank(cur[0]); cur[0].m = 0;
    for(int i = 1; i < w1; ++i) { blank(cur[i]); fromLeft(cur[i], at(0, i), cur[i - 1], endExt); }
    flip();
    int j = 1;
    auto leftAnchoredRow = [&](int row, bool lastRow) {
        blank(cur[0]);
        fromAbove(cur[0], at(row, 0), prev[0], endExt);
        const int stop = (row + w1 <= n1 + 1) ? (row + w1 - 1) : n1;
        int i = 1;
        for(; i != stop; ++i) {
            fromDiag(cur[i], at(row, i), prev[i - 1], sub(y[row], x[i]));
            fromAbove(cur[i], at(row, i), prev[i], ext);
            fromLeft(cur[i], at(row, i), cur[i - 1], lastRow ? endExt : ext);
        }
        fromDiag(cur[i], at(row, i), prev[i - 1], sub(y[row], x[i]));
        fromLeft(cur[i], at(row, i), cur[i - 1], lastRow ? endExt : ext);
        if(row + w1 - 1 > n1) fromAbove(cur[i], at(row, i), prev[i], endExt); else cur[i].i = kInf;
        flip();
    };
    const int anchored = w2 < n2 ? w2 : n2 - 1;
    for(; j <= anchored; ++j) leftAnchoredRow(j, false);
    if(j == n2 && w2 != n2 - 1) { leftAnchoredRow(j, true); ++j; }
    for(; j <= n2 - w2 + 1; ++j) {                                              // both band edges inside the row
        blank(cur[j - w2]);
        const int stop = j + w1 - 1;
        int i = j - w2 + 1;
        for(; i != stop; ++i) {
            fromDiag(cur[i], at(j, i), prev[i - 1], sub(y[j], x[i]));
            fromAbove(cur[i], at(j, i), prev[i], ext);
            fromLeft(cur[i], at(j, i), cur[i - 1], ext);
        }
        fromDiag(cur[i], at(j, i), prev[i - 1], sub(y[j], x[i]));
        fromLeft(cur[i], at(j, i), cur[i - 1], ext);
        cur[i].i = kInf;
        flip();
    }
    auto rightAnchoredRow = [&](int row, bool lastRow) {
        blank(cur[row - w2]);
        int i = row - w2 + 1;
        for(; i < n1; ++i) {
            fromDiag(cur[i], at(row, i), prev[i - 1], sub(y[row], x[i]));
            fromAbove(cur[i], at(row, i), prev[i], ext);
            fromLeft(cur[i], at(row, i), cur[i - 1], lastRow ? endExt : ext);
        }
        fromDiag(cur[i], at(row, i), prev[n1 - 1], sub(y[row], x[i]));
        fromAbove(cur[i], at(row, i), prev[i], endExt);
        fromLeft(cur[i], at(row, i), cur[i - 1], lastRow ? endExt : ext);
        flip();
    };
    for(; j < n2; ++j) rightAnchoredRow(j, false);
    if(j == n2) rightAnchoredRow(j, true);

    // trace back from the corner: state = the layer the current column sits in, next = the layer its predecessor sits in
    int i = n1;
    j = n2;
    const Cell& corner = prev[n1];
    int best = corner.m, state = 0, next = at(j, i) & 3;
    if(corner.i > best) { best = corner.i; state = 1; next = (at(j, i) >> 2) & 3; }
    if(corner.d > best) { best = corner.d; state = 2; next = (at(j, i) >> 4) & 3; }
    out.score = best;
    while(true) {
        ++out.columns;
        if(state == 0 && x[i] == y[j]) ++out.matches;
        if(state == 0) { --i; --j; } else if(state == 1) --j; else --i;
        if(i == 0 && j == 0) break;
        state = next;
        const uint8_t bp = at(j, i);
        next = state == 0 ? (bp & 3) : state == 1 ? ((bp >> 2) & 3) : ((bp >> 4) & 3);
    }
    return out;
}

} // namespace stride
