// oracle/fm_oracle.cpp -- TEST INFRASTRUCTURE ONLY (see fm_oracle.hpp).
#include "fm_oracle.hpp"

#include <algorithm>
#include <cassert>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <numeric>

namespace lrsc_oracle {

// ---------------------------------------------------------------------------------------
// Alphabet helpers
// ---------------------------------------------------------------------------------------
int bwt_rank_of(char b)
{
    // BWT_ALPHABET::getRank LUT (Util/Alphabet.h:87-111): A=1 C=2 G=3 T=4, everything else 0
    switch(b) {
        case 'A': return 1;
        case 'C': return 2;
        case 'G': return 3;
        case 'T': return 4;
        default: return 0;
    }
}
char bwt_char_of(int rank) { return "$ACGT"[rank]; }   // RANK_ALPHABET (Util/Alphabet.h:39)

char complement_base(char b)
{
    switch(b) {                                          // Util/Util.h:268-286
        case 'A': return 'T';
        case 'C': return 'G';
        case 'G': return 'C';
        case 'T': return 'A';
        case 'N': return 'N';
        default: assert(false && "complement of non-ACGTN"); return 'N';
    }
}
std::string reverse_str(const std::string& s) { return std::string(s.rbegin(), s.rend()); }
std::string reverse_complement(const std::string& s)
{
    std::string out(s.size(), 'A');
    const size_t n = s.size();
    for(size_t i = 0; i < n; ++i) out[i] = complement_base(s[n - 1 - i]);
    return out;
}

static inline int dna_idx(char b)
{
    // DNA_ALPHABET::getIdx via _ALPHABET (Util/Alphabet.cpp:15): A0 C1 G2 T3 $4
    switch(b) {
        case 'A': return 0;
        case 'C': return 1;
        case 'G': return 2;
        case 'T': return 3;
        default: return 4;
    }
}

// RLUnit (SuffixTools/RLUnit.h:13-16): data = (rank << 5) | run_len, run_len 1..31
static inline int unit_rank(uint8_t u) { return u >> 5; }
static inline unsigned unit_count(uint8_t u) { return u & 0x1F; }

// ---------------------------------------------------------------------------------------
// RLBwt
// ---------------------------------------------------------------------------------------
bool RLBwt::load(const std::string& path, std::string* err)
{
    std::ifstream in(path, std::ios::binary);
    if(!in) { if(err) *err = "cannot open " + path; return false; }
    uint16_t magic = 0;
    uint64_t nstr = 0, nsym = 0, nruns = 0;
    int32_t flag = 0;
    in.read(reinterpret_cast<char*>(&magic), 2);
    if(!in || magic != 0xCACA) {                     // BWTReader.h:33, BWTReaderBinary.cpp:61-65
        if(err) *err = "BWT file is not properly formatted, aborting";
        return false;
    }
    in.read(reinterpret_cast<char*>(&nstr), 8);
    in.read(reinterpret_cast<char*>(&nsym), 8);
    in.read(reinterpret_cast<char*>(&nruns), 8);
    in.read(reinterpret_cast<char*>(&flag), 4);
    if(!in || nruns == 0) { if(err) *err = "truncated BWT header"; return false; }
    std::vector<uint8_t> rl(nruns);
    in.read(reinterpret_cast<char*>(rl.data()), (std::streamsize)nruns);
    if(!in) { if(err) *err = "truncated BWT body"; return false; }
    assign(std::move(rl), nstr, nsym);
    return true;
}

void RLBwt::assign(std::vector<uint8_t> rl_units, uint64_t num_strings, uint64_t num_symbols)
{
    rl_ = std::move(rl_units);
    num_strings_ = num_strings;
    num_symbols_ = num_symbols;
    initialize_fm_index();
}

static uint64_t num_required_markers(uint64_t n, uint64_t d)   // RLBWT.cpp:251-257
{
    return (n % d == 0) ? (n / d) + 1 : (n / d) + 2;
}

void RLBwt::initialize_fm_index()
{
    small_shift_ = 5;    // Occurrence::calculateShiftValue(32)
    large_shift_ = 13;   // Occurrence::calculateShiftValue(8192)
    const uint64_t n_large = num_required_markers(num_symbols_, kLargeRate);
    const uint64_t n_small = num_required_markers(num_symbols_, kSmallRate);
    large_.assign(n_large, LargeMarker());
    small_.assign(n_small, SmallMarker());

    uint64_t curr_large = 1, curr_small = 1;
    uint64_t next_small = kSmallRate, next_large = kLargeRate;
    uint64_t running_total = 0;
    uint64_t running_ac[5] = {0, 0, 0, 0, 0};

    const uint64_t nunits = rl_.size();
    for(uint64_t i = 0; i < nunits; ++i) {
        const uint8_t u = rl_[i];
        const unsigned run_len = unit_count(u);
        running_ac[unit_rank(u) > 4 ? 0 : unit_rank(u)] += run_len;
        running_total += run_len;

        const uint64_t curr_unit_index = i + 1;
        const bool last_symbol = (i == nunits - 1);

        // large markers: placed AFTER the run crossing the boundary ends (RLBWT.cpp:160-181)
        bool place_last_large = last_symbol && curr_large < n_large;
        while(running_total >= next_large || place_last_large) {
            LargeMarker& m = large_[curr_large];
            m.unit_index = i + 1;
            for(int j = 0; j < 5; ++j) m.counts[j] = running_ac[j];
            next_large += kLargeRate;
            curr_large += 1;
            place_last_large = last_symbol && curr_large < n_large;
        }

        // small markers (RLBWT.cpp:184-236)
        bool place_last_small = last_symbol && curr_small < n_small;
        while(running_total >= next_small || place_last_small) {
            const uint64_t expected_marker_pos = curr_small * kSmallRate;
            const uint64_t large_idx = expected_marker_pos >> large_shift_;
            assert(large_idx < curr_large);
            const LargeMarker& prev_large = large_[large_idx];
            SmallMarker& sm = small_[curr_small];
            for(int j = 0; j < 5; ++j) {
                const uint64_t v = running_ac[j] - prev_large.counts[j];
                assert(v <= 0xFFFF);
                sm.counts[j] = (uint16_t)v;
            }
            sm.unit_count = (uint16_t)(curr_unit_index - prev_large.unit_index);
            next_small += kSmallRate;
            curr_small += 1;
            place_last_small = last_symbol && curr_small < n_small;
        }
    }
    assert(curr_small == n_small);
    assert(curr_large == n_large);

    // C(a) (RLBWT.cpp:243-247)
    pred_[0] = 0;
    pred_[1] = running_ac[0];
    pred_[2] = pred_[1] + running_ac[1];
    pred_[3] = pred_[2] + running_ac[2];
    pred_[4] = pred_[3] + running_ac[3];
}

LargeMarker RLBwt::interpolated_marker(uint64_t small_idx) const
{
    const uint64_t target_position = small_idx << small_shift_;
    const uint64_t large_idx = target_position >> large_shift_;
    LargeMarker m = large_[large_idx];
    const SmallMarker& rel = small_[small_idx];
    for(int j = 0; j < 5; ++j) m.counts[j] += rel.counts[j];
    m.unit_index += rel.unit_count;
    return m;
}

uint64_t RLBwt::nearest_marker_idx(uint64_t pos) const
{
    const uint64_t offset = pos & (kSmallRate - 1);
    const uint64_t base = pos >> small_shift_;
    return (offset < (uint64_t)(kSmallRate >> 1)) ? base : base + 1;
}

uint64_t& RLBwt::occ_calls_tls()
{
    static thread_local uint64_t n = 0;
    return n;
}

uint64_t RLBwt::occ(int rank, int64_t idx_signed) const
{
    ++occ_calls_tls();
    uint64_t idx = (uint64_t)idx_signed;
    ++idx;                                           // marker counts are exclusive (RLBWT.h:125)
    const LargeMarker marker = interpolated_marker(nearest_marker_idx(idx));
    uint64_t current_position = marker.actual_position();
    const bool forwards = current_position < idx;
    uint64_t running = marker.counts[rank];
    uint64_t symbol_index = marker.unit_index;
    if(forwards) {                                   // accumulateForwards RLBWT.h:217-230
        while(current_position != idx) {
            const uint64_t diff = idx - current_position;
            const uint8_t u = rl_[symbol_index];
            uint64_t run_len = unit_count(u);
            if(run_len > diff) run_len = diff;
            if(unit_rank(u) == rank) running += run_len;
            current_position += run_len;
            ++symbol_index;
        }
    } else {                                         // accumulateBackwards RLBWT.h:200-214
        while(current_position != idx) {
            const uint64_t diff = current_position - idx;
            --symbol_index;
            const uint8_t u = rl_[symbol_index];
            uint64_t run_len = unit_count(u);
            if(run_len > diff) run_len = diff;
            if(unit_rank(u) == rank) running -= run_len;
            current_position -= run_len;
        }
    }
    return running;
}

char RLBwt::get_char(uint64_t idx) const
{
    const LargeMarker upper = interpolated_marker((idx >> small_shift_) + 1);   // getUpperMarker
    uint64_t current_position = upper.actual_position();
    assert(current_position >= idx);
    uint64_t symbol_index = upper.unit_index;
    while(current_position > idx) {
        assert(symbol_index != 0);
        symbol_index -= 1;
        current_position -= unit_count(rl_[symbol_index]);
    }
    return bwt_char_of(unit_rank(rl_[symbol_index]));
}

std::string RLBwt::decode() const
{
    std::string out;
    out.reserve(num_symbols_);
    for(uint8_t u : rl_) out.append(unit_count(u), bwt_char_of(unit_rank(u)));
    return out;
}

void RLBwt::init_interval(Interval& iv, char b) const
{
    const int r = bwt_rank_of(b);
    iv.lower = (int64_t)pc(r);
    iv.upper = iv.lower + (int64_t)occ(r, (int64_t)num_symbols_ - 1) - 1;
}

void RLBwt::update_interval(Interval& iv, char b) const
{
    const int r = bwt_rank_of(b);
    const uint64_t pb = pc(r);
    iv.lower = (int64_t)(pb + occ(r, iv.lower - 1));
    iv.upper = (int64_t)(pb + occ(r, iv.upper) - 1);
}

Interval RLBwt::find_interval(const std::string& w, int* count) const
{
    const int len = (int)w.size();
    int j = len - 1;
    char curr = w[j];
    if(count != nullptr) count[dna_idx(curr)]++;
    Interval iv;
    init_interval(iv, curr);
    --j;
    for(; j >= 0; --j) {
        curr = w[j];
        if(count != nullptr) count[dna_idx(curr)]++;
        update_interval(iv, curr);
        if(!iv.valid()) break;
    }
    return iv;
}

BiInterval find_bi_interval(const IndexSet& idx, const std::string& w, int* count)
{
    BiInterval bi;
    bi.fwd = idx.rbwt->find_interval(reverse_str(w), count);
    bi.rvc = idx.bwt->find_interval(reverse_complement(w));
    return bi;
}

void update_bi_interval(BiInterval& bi, char b, const IndexSet& idx, int* count)
{
    if(count != nullptr) count[dna_idx(b)]++;
    idx.rbwt->update_interval(bi.fwd, b);
    idx.bwt->update_interval(bi.rvc, complement_base(b));
}

int64_t count_sequence_occurrences(const std::string& w, const RLBwt* bwt)
{
    BiInterval bi;
    bi.fwd = bwt->find_interval(w);
    bi.rvc = bwt->find_interval(reverse_complement(w));
    return bi.freq();
}

// ---------------------------------------------------------------------------------------
// Index construction by direct suffix sorting
// ---------------------------------------------------------------------------------------
std::string build_bwt_naive(const std::vector<std::string>& reads_in, bool reverse_reads)
{
    std::vector<std::string> rev;
    const std::vector<std::string>* reads = &reads_in;
    if(reverse_reads) {
        rev.reserve(reads_in.size());
        for(const auto& r : reads_in) rev.push_back(reverse_str(r));
        reads = &rev;
    }
    struct Suf { uint32_t read; uint32_t pos; };
    uint64_t total = 0;
    for(const auto& r : *reads) total += r.size() + 1;
    std::vector<Suf> sa;
    sa.reserve(total);
    for(uint32_t r = 0; r < reads->size(); ++r)
        for(uint32_t p = 0; p <= (*reads)[r].size(); ++p) sa.push_back({r, p});

    const std::vector<std::string>& R = *reads;
    std::sort(sa.begin(), sa.end(), [&R](const Suf& a, const Suf& b) {
        const std::string& sa_ = R[a.read];
        const std::string& sb_ = R[b.read];
        const size_t la = sa_.size() - a.pos, lb = sb_.size() - b.pos;
        const size_t m = la < lb ? la : lb;
        const int c = m ? std::memcmp(sa_.data() + a.pos, sb_.data() + b.pos, m) : 0;
        if(c != 0) return c < 0;
        if(la != lb) return la < lb;           // '$' sorts before A,C,G,T
        return a.read < b.read;                // sentinels in input order (MR_SO_IO)
    });

    std::string bwt(total, '$');
    for(uint64_t i = 0; i < total; ++i) {
        const Suf& s = sa[i];
        bwt[i] = (s.pos == 0) ? '$' : R[s.read][s.pos - 1];
    }
    return bwt;
}

std::vector<uint8_t> rl_encode(const std::string& bwt)
{
    std::vector<uint8_t> out;
    uint8_t cur = 0;   // uninitialised run == 0 (RLUnit.h:24,37)
    for(char b : bwt) {
        const uint8_t code = (uint8_t)(bwt_rank_of(b) << 5);
        if(cur != 0 && (cur & 0xE0) == code && (cur & 0x1F) != 31) {
            ++cur;
        } else {
            if(cur != 0) out.push_back(cur);
            cur = (uint8_t)(code | 1);
        }
    }
    if(cur != 0) out.push_back(cur);
    return out;
}

bool write_bwt_file(const std::string& path, uint64_t num_strings, uint64_t num_symbols,
                    const std::vector<uint8_t>& units)
{
    std::ofstream out(path, std::ios::binary);
    if(!out) return false;
    const uint16_t magic = 0xCACA;
    const uint64_t nruns = units.size();
    const int32_t flag = 0;   // BWF_NOFMI
    out.write(reinterpret_cast<const char*>(&magic), 2);
    out.write(reinterpret_cast<const char*>(&num_strings), 8);
    out.write(reinterpret_cast<const char*>(&num_symbols), 8);
    out.write(reinterpret_cast<const char*>(&nruns), 8);
    out.write(reinterpret_cast<const char*>(&flag), 4);
    out.write(reinterpret_cast<const char*>(units.data()), (std::streamsize)units.size());
    return (bool)out;
}

} // namespace lrsc_oracle
