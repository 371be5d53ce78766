// oracle/stdaln_oracle.cpp -- TEST INFRASTRUCTURE ONLY.  See stdaln_oracle.hpp.
#include "stdaln_oracle.hpp"

#include <cstdint>
#include <vector>

namespace lrsc_oracle {

namespace {

constexpr int kMinorInf = -1073741823;      // MINOR_INF (stdaln.h:84)
enum : uint8_t { kFromM = 0, kFromI = 1, kFromD = 2 };

struct Score { int M, I, D; };
struct Trace { uint8_t Mt, It, Dt; };

inline int nt4(char c)                      // aln_nt4_table (stdaln.c:54-71): A 0, G 1, C 2, T 3, anything else 4
{
    switch(c) {
        case 'A': case 'a': return 0;
        case 'G': case 'g': return 1;
        case 'C': case 'c': return 2;
        case 'T': case 't': return 3;
        default: return 4;
    }
}
inline int pacbio_score(int a, int b)       // aln_sm_pacbio (stdaln.c:231-237)
{
    if(a == 4 || b == 4) return -2;
    return a == b ? 1 : -8;
}

} // namespace

StdalnGlobal stdaln_global_pacbio(const std::string& str1, const std::string& str2)
{
    const int gap_open = 1, gap_ext = 1, gap_end = 0, band = 50;       // aln_param_pacbio (stdaln.c:248)
    const int len1 = (int)str1.size(), len2 = (int)str2.size();
    StdalnGlobal out{0, 0, 0};
    if(len1 == 0 || len2 == 0) return out;
    std::vector<uint8_t> seq1(len1 + 1), seq2(len2 + 1);                // 1-based like the reference after --seq1; --seq2
    for(int i = 0; i < len1; ++i) seq1[i + 1] = (uint8_t)nt4(str1[i]);
    for(int j = 0; j < len2; ++j) seq2[j + 1] = (uint8_t)nt4(str2[j]);

    int b1, b2;
    if(len1 > len2) { b1 = len1 - len2 + band; b2 = band; }
    else { b1 = band; b2 = len2 - len1 + band; }
    if(b1 > len1) b1 = len1;
    if(b2 > len2) b2 = len2;

    // trace of every cell (the reference allocates only the band of each row and shifts the row pointers; a dense matrix
    // addressed [j][i] holds the same cells)
    std::vector<Trace> tr((size_t)(len2 + 1) * (len1 + 1), Trace{0, 0, 0});
    auto cell = [&](int j, int i) -> Trace& { return tr[(size_t)j * (len1 + 1) + i]; };
    std::vector<Score> rowA(len1 + 1, Score{0, 0, 0}), rowB(len1 + 1, Score{0, 0, 0});
    Score* curr = rowA.data();
    Score* last = rowB.data();

    auto set_M = [&](Score& s, Trace& c, const Score& p, int sc) {
        if(p.M >= p.I) {
            if(p.M >= p.D) { s.M = p.M + sc; c.Mt = kFromM; } else { s.M = p.D + sc; c.Mt = kFromD; }
        } else {
            if(p.I > p.D) { s.M = p.I + sc; c.Mt = kFromI; } else { s.M = p.D + sc; c.Mt = kFromD; }
        }
    };
    auto set_I = [&](Score& s, Trace& c, const Score& p, int ext) {
        if(p.M - gap_open > p.I) { c.It = kFromM; s.I = p.M - gap_open - ext; } else { c.It = kFromI; s.I = p.I - ext; }
    };
    auto set_D = [&](Score& s, Trace& c, const Score& p, int ext) {
        if(p.M - gap_open > p.D) { c.Dt = kFromM; s.D = p.M - gap_open - ext; } else { c.Dt = kFromD; s.D = p.D - ext; }
    };
    // gap_end >= 0 here, so the "end" forms always price an extension with gap_end
    const int end_ext = gap_end;
    auto set_inf = [&](Score& s) { s.M = s.I = s.D = kMinorInf; };
    auto swap_rows = [&]() { Score* t = curr; curr = last; last = t; };

    // first row
    set_inf(curr[0]); curr[0].M = 0;
    for(int i = 1; i < b1; ++i) { set_inf(curr[i]); set_D(curr[i], cell(0, i), curr[i - 1], end_ext); }
    swap_rows();

    int j;
    // part 1: rows whose band starts at column 0
    const int tmp_end = b2 < len2 ? b2 : len2 - 1;
    auto row_from_zero = [&](int jj, bool last_row) {
        set_inf(curr[0]);
        set_I(curr[0], cell(jj, 0), last[0], end_ext);
        const int end = (jj + b1 <= len1 + 1) ? (jj + b1 - 1) : len1;
        int i = 1;
        for(; i != end; ++i) {
            set_M(curr[i], cell(jj, i), last[i - 1], pacbio_score(seq2[jj], seq1[i]));
            set_I(curr[i], cell(jj, i), last[i], gap_ext);
            set_D(curr[i], cell(jj, i), curr[i - 1], last_row ? end_ext : gap_ext);
        }
        set_M(curr[i], cell(jj, i), last[i - 1], pacbio_score(seq2[jj], seq1[i]));
        set_D(curr[i], cell(jj, i), curr[i - 1], last_row ? end_ext : gap_ext);
        if(jj + b1 - 1 > len1) set_I(curr[i], cell(jj, i), last[i], end_ext);
        else curr[i].I = kMinorInf;
        swap_rows();
    };
    for(j = 1; j <= tmp_end; ++j) row_from_zero(j, false);
    if(j == len2 && b2 != len2 - 1) { row_from_zero(j, true); ++j; }

    // part 2: the band slides, its right edge is still inside the row
    for(; j <= len2 - b2 + 1; ++j) {
        set_inf(curr[j - b2]);
        const int end = j + b1 - 1;
        int i = j - b2 + 1;
        for(; i != end; ++i) {
            set_M(curr[i], cell(j, i), last[i - 1], pacbio_score(seq2[j], seq1[i]));
            set_I(curr[i], cell(j, i), last[i], gap_ext);
            set_D(curr[i], cell(j, i), curr[i - 1], gap_ext);
        }
        set_M(curr[i], cell(j, i), last[i - 1], pacbio_score(seq2[j], seq1[i]));
        set_D(curr[i], cell(j, i), curr[i - 1], gap_ext);
        curr[i].I = kMinorInf;
        swap_rows();
    }
    // part 3: the band's right edge is the last column
    auto row_to_end = [&](int jj, bool last_row) {
        set_inf(curr[jj - b2]);
        int i = jj - b2 + 1;
        for(; i < len1; ++i) {
            set_M(curr[i], cell(jj, i), last[i - 1], pacbio_score(seq2[jj], seq1[i]));
            set_I(curr[i], cell(jj, i), last[i], gap_ext);
            set_D(curr[i], cell(jj, i), curr[i - 1], last_row ? end_ext : gap_ext);
        }
        set_M(curr[i], cell(jj, i), last[len1 - 1], pacbio_score(seq2[jj], seq1[i]));
        set_I(curr[i], cell(jj, i), last[i], end_ext);
        set_D(curr[i], cell(jj, i), curr[i - 1], last_row ? end_ext : gap_ext);
        swap_rows();
    };
    for(; j < len2; ++j) row_to_end(j, false);
    if(j == len2) row_to_end(j, true);

    // backtrace
    int i = len1;
    j = len2;
    const Score& fin = last[len1];
    int max = fin.M;
    uint8_t type = cell(j, i).Mt, ctype = kFromM;
    if(fin.I > max) { max = fin.I; type = cell(j, i).It; ctype = kFromI; }
    if(fin.D > max) { max = fin.D; type = cell(j, i).Dt; ctype = kFromD; }
    out.score = max;
    // the path is emitted from its end; outm counts the FROM_M steps over equal, non-N nucleotides (stdaln.c:833-837)
    int path_len = 0, matches = 0;
    auto emit = [&](uint8_t ct, int pi, int pj) {
        if(ct == kFromM && pi >= 1 && pj >= 1 && seq1[pi] == seq2[pj] && seq1[pi] != 5) ++matches;
        ++path_len;
    };
    emit(ctype, i, j);
    do {
        switch(ctype) {
            case kFromM: --i; --j; break;
            case kFromI: --j; break;
            default: --i; break;
        }
        const Trace& q = cell(j, i);
        ctype = type;
        switch(type) {
            case kFromM: type = q.Mt; break;
            case kFromI: type = q.It; break;
            default: type = q.Dt; break;
        }
        if(i || j) emit(ctype, i, j);      // the element recorded at (0, 0) is beyond path_len (stdaln.c:534: path_len = p - path - 1)
    } while(i || j);
    out.path_len = path_len;
    out.matches = matches;
    return out;
}

} // namespace lrsc_oracle
