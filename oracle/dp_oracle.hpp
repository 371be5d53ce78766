// oracle/dp_oracle.hpp -- TEST INFRASTRUCTURE ONLY.
//
// CPU restatement of the DP/MSA fallback that `pbcorrect` runs when FM-extend fails
// (PacBio/PacBioSelfCorrectionProcess.cpp:208-245): LongReadOverlap::buildMultipleAlignment /
// retrieveMatches / retrieveStr (PacBio/LongReadOverlap.cpp:17-55,593-756), Overlapper::extendMatch
// (Thirdparty/overlapper.cpp:421-701) and MultipleAlignment::addOverlap / calculateBaseConsensus
// (Thirdparty/multiple_alignment.cpp:208-393,517-594).
//
// Parity pin: extend_match is checked against the reference's own overlapper.cpp object code
// (oracle/_ref).  multiple_alignment.cpp and LongReadOverlap.cpp include Util/HashMap.h and cannot
// be built here ("parity unpinned" by a reference build for those two; line-by-line restatement).
#pragma once
#include <string>
#include <vector>

#include "fm_oracle.hpp"

namespace lrsc_oracle {

struct SequenceInterval { int start = 0, end = -1; };                  // overlapper.cpp:50-53
struct SequenceOverlap {                                               // overlapper.h:69-125
    SequenceInterval match[2];
    int length[2] = {0, 0};
    int score = -1;
    int edit_distance = -1;
    int total_columns = -1;
    std::string cigar;
    double getPercentIdentity() const { return (double)(total_columns - edit_distance) * 100.0f / total_columns; }   // :71-74
    int getOverlapLength() const { return total_columns; }
};

SequenceOverlap extend_match(const std::string& s1, const std::string& s2, int start_1, int start_2, int band_width,
                             const int MATCH_SCORE, const int GAP_PENALTY, const int MISMATCH_PENALTY);   // overlapper.cpp:421-701
std::string compact_cigar(const std::string& ecigar);                                                      // :1506-1527

struct MultipleAlignmentElement {                                      // multiple_alignment.cpp:50-180, .h:36-90
    MultipleAlignmentElement(const std::string& n, const std::string& s, const std::string& q, size_t leading, size_t trailing)
        : name(n), padded_sequence(s), padded_quality(q), leading_columns(leading), trailing_columns(trailing) {}
    size_t getNumColumns() const { return leading_columns + padded_sequence.size() + trailing_columns; }
    char getColumnSymbol(size_t column_idx) const;
    size_t getStartColumn() const { return leading_columns; }
    size_t getEndColumn() const { return getNumColumns() - trailing_columns - 1; }
    int getPaddedPositionOfBase(size_t idx) const;
    void insertGapBeforeColumn(size_t column_index);
    std::string name, padded_sequence, padded_quality;
    size_t leading_columns, trailing_columns;
};

class MultipleAlignment {
public:
    void addBaseSequence(const std::string& name, const std::string& sequence, const std::string& quality);   // :208-214
    void addOverlap(const std::string& name, const std::string& sequence, const std::string& quality,
                    const SequenceOverlap& overlap);                                                            // :216-224
    std::string calculateBaseConsensus(int min_call_coverage, int min_trim_coverage);                           // :517-594
    size_t getNumRows() const { return m_sequences.size(); }                                                    // :987-991
    std::vector<MultipleAlignmentElement> m_sequences;
private:
    void _addSequence(const std::string& name, const std::string& sequence, const std::string& quality,
                      size_t template_element_index, const SequenceOverlap& overlap, bool is_extension);        // :240-393
    void insertGapBeforeColumn(size_t column_index);                                                            // :1205-1211
    std::vector<int> getColumnBaseCounts(size_t idx) const;                                                     // :1293-1305
    static std::string expandCigar(const std::string& cigar);                                                   // :1232-1242
    static int symbol2index(char symbol);                                                                       // :1244-1265
};

struct SequenceOverlapPair {                                            // Algorithm/KmerOverlaps.h:18-27
    std::string sequence[2];
    bool is_reversed = false;
    SequenceOverlap overlap;
};

// PacBio/LongReadOverlap.cpp
void retrieveStr(const std::string& query, size_t seedSize, size_t maxLength, const IndexSet& indices, bool isRC,
                 size_t coverage, std::vector<std::string>& ovlStr);                                            // :667-756
void retrieveMatches(const std::string& query, size_t k, size_t min_overlap, double min_identity, size_t coverage,
                     const IndexSet& indices, bool isRC, std::vector<SequenceOverlapPair>& overlap_vector);     // :593-662
MultipleAlignment buildMultipleAlignment(const std::string& query, size_t srcKmerLength, size_t tarKmerLength,
                                         size_t min_overlap, double min_identity, size_t coverage,
                                         const IndexSet& indices);                                              // :17-55

} // namespace lrsc_oracle
