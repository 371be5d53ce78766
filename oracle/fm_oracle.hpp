// oracle/fm_oracle.hpp -- TEST INFRASTRUCTURE ONLY.
//
// CPU restatement of the reference's FM-index layer (run-length BWT + two-level
// occurrence markers + backward search).  Only tests/, __graft_entry__.smoke()
// and bench.py's cpu_baseline leg may link or load this; the product
// (longreadselfcorrect_amd/csrc) never does.
//
// Parity pin: every function here is checked against the reference's own object
// code (oracle/_ref/liblrsc_ref.so, built from /root/reference/SuffixTools/RLBWT.cpp
// etc.) by tests/test_oracle_vs_ref.py, and against the fixtures that run
// committed under tests/golden/.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace lrsc_oracle {

// Rank alphabet "$ACGT" = 0..4 (Util/Alphabet.h:39,87-111); every other byte ranks 0.
int bwt_rank_of(char b);
char bwt_char_of(int rank);
// complement(char) for ACGT (Util/Util.h:268-286)
char complement_base(char b);
std::string reverse_str(const std::string& s);             // Util/Util.cpp:43-46
std::string reverse_complement(const std::string& s);      // Util/Util.cpp:19-28

// BWTInterval (SuffixTools/BWTInterval.h:19-81)
struct Interval {
    int64_t lower = 0;
    int64_t upper = 0;
    bool valid() const { return lower <= upper; }
    int64_t size() const { return upper - lower + 1; }
    int64_t freq() const { return valid() ? size() : 0; }
};
// BiBWTInterval (SuffixTools/BWTInterval.h:82-100)
struct BiInterval {
    Interval fwd;  // interval of reverse(w) in the rBWT
    Interval rvc;  // interval of revcomp(w) in the BWT
    bool valid() const { return fwd.valid() && rvc.valid(); }
    int64_t freq() const { return fwd.freq() + rvc.freq(); }
};

// LargeMarker / SmallMarker (SuffixTools/FMMarkers.h:19-61, 68-93)
struct LargeMarker {
    uint64_t counts[5] = {0, 0, 0, 0, 0};
    uint64_t unit_index = 0;
    uint64_t actual_position() const { return counts[0] + counts[1] + counts[2] + counts[3] + counts[4]; }
};
struct SmallMarker {
    uint16_t counts[5] = {0, 0, 0, 0, 0};
    uint16_t unit_count = 0;
};

// RLBWT (SuffixTools/RLBWT.h:27-298, RLBWT.cpp:23-32,109-257)
class RLBwt {
public:
    static constexpr int kSmallRate = 32;    // RLBWT.h:265
    static constexpr int kLargeRate = 8192;  // RLBWT.h:264

    // Parse a binary .bwt/.rbwt (BWTReaderBinary.cpp:26-85): returns false + message on error.
    bool load(const std::string& path, std::string* err);
    // Take an RL-unit string directly (used by the builder and tests).
    void assign(std::vector<uint8_t> rl_units, uint64_t num_strings, uint64_t num_symbols);

    uint64_t num_strings() const { return num_strings_; }
    uint64_t num_symbols() const { return num_symbols_; }
    uint64_t num_runs() const { return rl_.size(); }
    const std::vector<uint8_t>& units() const { return rl_; }

    uint64_t pc(int rank) const { return pred_[rank]; }                 // RLBWT.h:118
    // #symbols of rank `rank` in bwt[0..idx], idx may be -1 (RLBWT.h:121-140)
    uint64_t occ(int rank, int64_t idx) const;
    char get_char(uint64_t idx) const;                                  // RLBWT.h:42-63
    // decode to one byte per symbol (test helper, not in the reference)
    std::string decode() const;

    // Backward search (BWTAlgorithms.h:66-72,136-140; BWTAlgorithms.cpp:14-31)
    void init_interval(Interval& iv, char b) const;
    void update_interval(Interval& iv, char b) const;
    // `count` (optional, int[4] indexed A,C,G,T) follows BWTAlgorithms.cpp:19 / .h:68
    Interval find_interval(const std::string& w, int* count = nullptr) const;

    // counter of occ() calls made by the calling thread, summed over all RLBwt objects (cpu-baseline accounting; not in the
    // reference).  Thread-local: bench.py's cpu_baseline runs one oracle thread per host core over a shared index.
    static uint64_t& occ_calls_tls();

private:
    void initialize_fm_index();                                          // RLBWT.cpp:109-248
    LargeMarker interpolated_marker(uint64_t small_idx) const;           // RLBWT.h:105-116
    uint64_t nearest_marker_idx(uint64_t pos) const;                     // RLBWT.h:66-79

    std::vector<uint8_t> rl_;
    std::vector<LargeMarker> large_;
    std::vector<SmallMarker> small_;
    uint64_t pred_[5] = {0, 0, 0, 0, 0};
    uint64_t num_strings_ = 0, num_symbols_ = 0;
    int small_shift_ = 5, large_shift_ = 13;
};

struct IndexSet {       // BWTIndexSet.h:23-34 (pBWT, pRBWT only)
    const RLBwt* bwt = nullptr;
    const RLBwt* rbwt = nullptr;
};

// BWTAlgorithms.cpp:32-38 ; :135-141 ; BWTAlgorithms.h:73-77
BiInterval find_bi_interval(const IndexSet& idx, const std::string& w, int* count = nullptr);
void update_bi_interval(BiInterval& bi, char b, const IndexSet& idx, int* count = nullptr);
int64_t count_sequence_occurrences(const std::string& w, const RLBwt* bwt);

// ---- index construction (restates the *result* of `stride index -a ropebwt2`) ----
// Multi-string BWT with sentinels ordered by input order (BWTCARopebwt.cpp:167, MR_SO_IO),
// '$' < A < C < G < T, by direct suffix sorting.  reverse_reads=true gives the .rbwt.
// Returns one byte per row.
std::string build_bwt_naive(const std::vector<std::string>& reads, bool reverse_reads);
// RL-encode exactly as BWTWriterBinary::writeBWChar (BWTWriterBinary.cpp:50-71).
std::vector<uint8_t> rl_encode(const std::string& bwt);
// 30-byte header + units (BWTWriterBinary.cpp:28-46,82-93).
bool write_bwt_file(const std::string& path, uint64_t num_strings, uint64_t num_symbols,
                    const std::vector<uint8_t>& units);

} // namespace lrsc_oracle
