// oracle/dp_oracle.cpp -- TEST INFRASTRUCTURE ONLY (see dp_oracle.hpp).
#include "dp_oracle.hpp"

#include <algorithm>
#include <cassert>
#include <cctype>
#include <cstdlib>
#include <iostream>
#include <limits>
#include <sstream>

namespace lrsc_oracle {

typedef std::vector<int> DPCells;

// =======================================================================================
// Overlapper (Thirdparty/overlapper.cpp)
// =======================================================================================
static inline int _getBandedCellIndex(int i, int j, int band_width, int band_origin_row)   // :391-396
{
    int band_start = band_origin_row + i;
    int band_row_index = j - band_start;
    return (band_row_index >= 0 && band_row_index < band_width) ? i * band_width + band_row_index : -1;
}
static inline int _getBandedCellScore(const DPCells& cells, int i, int j, int band_width, int band_origin_row,
                                      int invalid_score)                                   // :399-404
{
    int band_start = band_origin_row + i;
    int band_row_index = j - band_start;
    return (band_row_index >= 0 && band_row_index < band_width) ? cells[i * band_width + band_row_index] : invalid_score;
}

std::string compact_cigar(const std::string& ecigar)
{
    if(ecigar.empty()) return "";
    std::stringstream compact_cigar;
    char curr_symbol = ecigar[0];
    int curr_run = 1;
    for(size_t i = 1; i < ecigar.size(); ++i) {
        if(ecigar[i] == curr_symbol) {
            curr_run += 1;
        } else {
            compact_cigar << curr_run << curr_symbol;
            curr_symbol = ecigar[i];
            curr_run = 1;
        }
    }
    compact_cigar << curr_run << curr_symbol;
    return compact_cigar.str();
}

#define max3(x, y, z) std::max(std::max(x, y), z)

SequenceOverlap extend_match(const std::string& s1, const std::string& s2, int start_1, int start_2, int band_width,
                             const int MATCH_SCORE, const int GAP_PENALTY, const int MISMATCH_PENALTY)
{
    SequenceOverlap output;
    int num_columns = s1.size() + 1;
    int num_rows = s2.size() + 1;

    // Calculate the number of cells off the diagonal to compute
    int half_width = band_width / 2;
    band_width = half_width * 2 + 1;   // the total number of cells per band

    size_t num_cells_required = num_columns * band_width;
    int INVALID_SCORE = std::numeric_limits<int>::min();
    DPCells cells(num_cells_required, 0);

    int band_center = start_2 - start_1 + 1;
    int band_origin = band_center - (half_width + 1);

    // Fill in the bands column by column
    for(int i = 1; i < num_columns; ++i) {
        int j = band_origin + i;   // start row of this band
        int end_row = j + band_width;

        // Trim band coordinates to only compute valid positions
        if(j < 1) j = 1;
        if(end_row > num_rows) end_row = num_rows;
        if(end_row <= 0 || j >= num_rows || j >= end_row) continue;   // nothing to do for this column

        int curr_idx = _getBandedCellIndex(i, j, band_width, band_origin);
        int left_idx = _getBandedCellIndex(i - 1, j, band_width, band_origin);
        int diagonal_idx = _getBandedCellIndex(i - 1, j - 1, band_width, band_origin);
        int diagonal_score = cells[diagonal_idx] + (s1[i - 1] == s2[j - 1] ? MATCH_SCORE : MISMATCH_PENALTY);
        int left_score = left_idx != -1 ? cells[left_idx] + GAP_PENALTY : INVALID_SCORE;
        int up_score = 0;

        // Set the first row score
        cells[curr_idx] = std::max(left_score, diagonal_score);

        curr_idx += 1;
        left_idx += 1;
        diagonal_idx += 1;
        j += 1;

        // Fill in the main part of the band, stopping before the last row
        while(j < end_row - 1) {
            diagonal_score = cells[diagonal_idx] + (s1[i - 1] == s2[j - 1] ? MATCH_SCORE : MISMATCH_PENALTY);
            left_score = cells[left_idx] + GAP_PENALTY;
            up_score = cells[curr_idx - 1] + GAP_PENALTY;
            cells[curr_idx] = max3(diagonal_score, left_score, up_score);
            curr_idx += 1;
            left_idx += 1;
            diagonal_idx += 1;
            j += 1;
        }

        // Fill in last row, here we ignore the left cell which is now out of band
        if(j != end_row) {
            diagonal_score = cells[diagonal_idx] + (s1[i - 1] == s2[j - 1] ? MATCH_SCORE : MISMATCH_PENALTY);
            up_score = cells[curr_idx - 1] + GAP_PENALTY;
            cells[curr_idx] = std::max(diagonal_score, up_score);
        }
    }

    int max_row_value = std::numeric_limits<int>::min();
    int max_column_value = std::numeric_limits<int>::min();
    size_t max_row_index = 0;
    size_t max_column_index = 0;

    // Check every column of the last row. The first column is skipped to avoid empty alignments
    for(int i = 1; i < num_columns; ++i) {
        int v = _getBandedCellScore(cells, i, num_rows - 1, band_width, band_origin, INVALID_SCORE);
        if(v > max_row_value) {
            max_row_value = v;
            max_row_index = i;
        }
    }
    // Check every row of the last column
    for(int j = 1; j < num_rows; ++j) {
        int v = _getBandedCellScore(cells, num_columns - 1, j, band_width, band_origin, INVALID_SCORE);
        if(v > max_column_value) {
            max_column_value = v;
            max_column_index = j;
        }
    }

    size_t i;
    size_t j;
    if(max_column_value > max_row_value) {
        i = num_columns - 1;
        j = max_column_index;
        output.score = max_column_value;
    } else {
        i = max_row_index;
        j = num_rows - 1;
        output.score = max_row_value;
    }

    output.match[0].end = i - 1;
    output.match[1].end = j - 1;
    output.length[0] = s1.length();
    output.length[1] = s2.length();
    output.edit_distance = 0;
    output.total_columns = 0;

    std::string cigar;
    while(i > 0 && j > 0) {
        int idx_1 = i - 1;
        int idx_2 = j - 1;

        bool is_match = s1[idx_1] == s2[idx_2];
        int diagonal = _getBandedCellScore(cells, i - 1, j - 1, band_width, band_origin, INVALID_SCORE) +
                       (is_match ? MATCH_SCORE : MISMATCH_PENALTY);
        int up = _getBandedCellScore(cells, i, j - 1, band_width, band_origin, INVALID_SCORE) + GAP_PENALTY;
        int left = _getBandedCellScore(cells, i - 1, j, band_width, band_origin, INVALID_SCORE) + GAP_PENALTY;
        int curr = _getBandedCellScore(cells, i, j, band_width, band_origin, INVALID_SCORE);

        // s2 homopolymer, prefer s2 extension (s2[j] may be the terminating NUL: overlapper.cpp:625)
        if(s2[idx_2] == s2[j]) {
            if(curr == up) {
                cigar.push_back('I');
                j -= 1;
                output.edit_distance += 1;
            } else if(curr == left) {
                cigar.push_back('D');
                i -= 1;
                output.edit_distance += 1;
            } else {
                assert(curr == diagonal);
                if(!is_match) output.edit_distance += 1;
                cigar.push_back('M');
                i -= 1;
                j -= 1;
            }
        }
        // s1 homopolymer, prefer s1 extension
        else if(s1[idx_1] == s1[i]) {
            if(curr == left) {
                cigar.push_back('D');
                i -= 1;
                output.edit_distance += 1;
            } else if(curr == up) {
                cigar.push_back('I');
                j -= 1;
                output.edit_distance += 1;
            } else {
                assert(curr == diagonal);
                if(!is_match) output.edit_distance += 1;
                cigar.push_back('M');
                i -= 1;
                j -= 1;
            }
        } else {
            if(curr == diagonal) {
                if(!is_match) output.edit_distance += 1;
                cigar.push_back('M');
                i -= 1;
                j -= 1;
            } else if(curr == left) {
                cigar.push_back('D');
                i -= 1;
                output.edit_distance += 1;
            } else {
                assert(curr == up);
                cigar.push_back('I');
                j -= 1;
                output.edit_distance += 1;
            }
        }
        output.total_columns += 1;
    }

    output.match[0].start = i;
    output.match[1].start = j;

    std::reverse(cigar.begin(), cigar.end());
    assert(!cigar.empty());
    output.cigar = compact_cigar(cigar);
    return output;
}

// =======================================================================================
// MultipleAlignment (Thirdparty/multiple_alignment.cpp)
// =======================================================================================
char MultipleAlignmentElement::getColumnSymbol(size_t column_idx) const
{
    assert(column_idx < getNumColumns());
    if(column_idx < leading_columns || column_idx >= leading_columns + padded_sequence.size()) {
        return '\0';
    } else {
        return padded_sequence[column_idx - leading_columns];
    }
}

int MultipleAlignmentElement::getPaddedPositionOfBase(size_t idx) const
{
    size_t unpadded_count = 0;
    for(size_t i = 0; i < padded_sequence.size(); ++i) {
        if(padded_sequence[i] != '-') {
            if(unpadded_count == idx)
                return i;
            else
                unpadded_count += 1;
        }
    }
    std::cerr << "Base index out of bounds: " << idx << "\n";
    assert(false);
    return -1;
}

void MultipleAlignmentElement::insertGapBeforeColumn(size_t column_index)
{
    size_t first_sequence_column_index = leading_columns;
    if(column_index <= first_sequence_column_index) {
        leading_columns += 1;
    } else {
        assert(column_index > leading_columns);
        size_t insert_position = column_index - leading_columns;
        if(insert_position < padded_sequence.size()) {
            padded_sequence.insert(insert_position, 1, '-');
            if(!padded_quality.empty()) padded_quality.insert(insert_position, 1, '-');
        } else
            trailing_columns += 1;
    }
}

void MultipleAlignment::addBaseSequence(const std::string& name, const std::string& sequence, const std::string& quality)
{
    m_sequences.push_back(MultipleAlignmentElement(name, sequence, quality, 0, 0));
}

void MultipleAlignment::addOverlap(const std::string& incoming_name, const std::string& incoming_sequence,
                                   const std::string& incoming_quality, const SequenceOverlap& reference_incoming_overlap)
{
    assert(!m_sequences.empty());
    _addSequence(incoming_name, incoming_sequence, incoming_quality, 0, reference_incoming_overlap, false);
}

void MultipleAlignment::_addSequence(const std::string& name, const std::string& sequence, const std::string& quality,
                                     size_t template_element_index, const SequenceOverlap& overlap, bool is_extension)
{
    MultipleAlignmentElement* template_element = &m_sequences[template_element_index];
    const std::string& template_padded = template_element->padded_sequence;

    std::string padded_output;
    std::string padded_quality;
    assert(quality.empty() || quality.size() == sequence.size());

    size_t cigar_index = 0;
    size_t template_index = template_element->getPaddedPositionOfBase(overlap.match[0].start);
    size_t incoming_index = overlap.match[1].start;
    if(is_extension) assert(incoming_index == 0);

    size_t template_leading = template_element->leading_columns;
    size_t incoming_leading = template_index + template_leading;

    std::string expanded_cigar = expandCigar(overlap.cigar);
    assert(!expanded_cigar.empty());
    assert(template_index < template_padded.size());
    assert(template_padded[template_index] != '-');

    while(cigar_index < expanded_cigar.size()) {
        // Check if we are in an existing template gap. This must be handled seperately
        bool in_template_gap = template_padded[template_index] == '-';
        if(in_template_gap) {
            if(expanded_cigar[cigar_index] == 'I') {
                padded_output.push_back(sequence[incoming_index]);
                if(!quality.empty()) padded_quality.push_back(quality[incoming_index]);
                incoming_index += 1;
                cigar_index += 1;
                template_index += 1;
            } else {
                padded_output.push_back('-');
                if(!quality.empty()) padded_quality.push_back('-');
                template_index += 1;
            }
        } else {
            switch(expanded_cigar[cigar_index]) {
                case 'M':
                    padded_output.push_back(sequence[incoming_index]);
                    if(!quality.empty()) padded_quality.push_back(quality[incoming_index]);
                    incoming_index += 1;
                    template_index += 1;
                    cigar_index += 1;
                    break;
                case 'I':
                    insertGapBeforeColumn(template_index + template_leading);
                    padded_output.push_back(sequence[incoming_index]);
                    if(!quality.empty()) padded_quality.push_back(quality[incoming_index]);
                    incoming_index += 1;
                    cigar_index += 1;
                    template_index += 1;   // skip the newly introduced gap
                    break;
                case 'D':
                    padded_output.push_back('-');
                    if(!quality.empty()) padded_quality.push_back('-');
                    cigar_index += 1;
                    template_index += 1;
                    break;
                case 'S':
                    cigar_index += 1;
                    break;
                default:
                    std::cerr << "Error: unhandled cigar symbol " << expanded_cigar[cigar_index] << "\n";
                    exit(EXIT_FAILURE);
                    break;
            }
        }
    }

    if(is_extension) {
        padded_output.append(sequence.substr(incoming_index));
        padded_quality.append(quality.substr(incoming_index));
        size_t incoming_columns = padded_output.size() + incoming_leading;
        assert(incoming_columns >= m_sequences.front().getNumColumns());
        (void)incoming_columns;
        for(size_t i = 0; i < m_sequences.size(); ++i) m_sequences[i].trailing_columns += (sequence.size() - incoming_index);
    }

    size_t incoming_trailing = template_element->getNumColumns() - padded_output.size() - incoming_leading;
    if(is_extension) assert(incoming_trailing == 0);

    MultipleAlignmentElement incoming_element(name, padded_output, padded_quality, incoming_leading, incoming_trailing);
    m_sequences.push_back(incoming_element);
}

void MultipleAlignment::insertGapBeforeColumn(size_t column_index)
{
    for(size_t i = 0; i < m_sequences.size(); ++i) m_sequences[i].insertGapBeforeColumn(column_index);
}

std::string MultipleAlignment::expandCigar(const std::string& cigar)
{
    std::string out;
    std::stringstream parser(cigar);
    int length;
    char symbol;
    while(parser >> length >> symbol) out.append(length, symbol);
    return out;
}

int MultipleAlignment::symbol2index(char symbol)
{
    switch(std::toupper(symbol)) {
        case 'A': return 0;
        case 'C': return 1;
        case 'G': return 2;
        case 'T': return 3;
        case '-': return 5;
        default: return 4;   // all ambiguity codes get index 4
    }
}

std::vector<int> MultipleAlignment::getColumnBaseCounts(size_t idx) const
{
    std::vector<int> out(6, 0);
    for(size_t i = 0; i < m_sequences.size(); ++i) {
        char symbol = m_sequences[i].getColumnSymbol(idx);
        if(symbol != '\0') out[symbol2index(symbol)] += 1;
    }
    return out;
}

std::string MultipleAlignment::calculateBaseConsensus(int min_call_coverage, int min_trim_coverage)
{
    static const char* m_alphabet = "ACGTN-";   // multiple_alignment.cpp:43
    assert(!m_sequences.empty());
    std::string consensus_sequence;
    MultipleAlignmentElement& base_element = m_sequences.front();
    size_t start_column = base_element.getStartColumn();
    size_t end_column = base_element.getEndColumn();
    int last_good_base = -1;

    for(size_t c = start_column; c <= end_column; ++c) {
        std::vector<int> counts = getColumnBaseCounts(c);

        char max_symbol = '\0';
        int max_count = -1;
        int total_depth = 0;
        for(size_t a = 0; a < 6; ++a) {
            char symbol = m_alphabet[a];
            total_depth += counts[a];
            if(symbol != 'N' && counts[a] > max_count) {
                max_symbol = symbol;
                max_count = counts[a];
            }
        }

        char base_symbol = base_element.getColumnSymbol(c);
        int base_count = counts[symbol2index(base_symbol)];

        char consensus_symbol;
        if(max_count >= base_count && base_count < min_call_coverage)
            consensus_symbol = max_symbol;
        else
            consensus_symbol = base_symbol;

        if(consensus_symbol != '-' && (!consensus_sequence.empty() || total_depth >= min_trim_coverage))
            consensus_sequence.push_back(consensus_symbol);

        if(total_depth >= min_trim_coverage) {
            int consensus_index = consensus_sequence.size() - 1;
            if(consensus_index > last_good_base) last_good_base = consensus_index;
        }
    }

    if(last_good_base != -1)
        consensus_sequence.erase(last_good_base + 1);
    else
        consensus_sequence.clear();
    return consensus_sequence;
}

// =======================================================================================
// LongReadOverlap (PacBio/LongReadOverlap.cpp)
// =======================================================================================
void retrieveStr(const std::string& query, size_t seedSize, size_t maxLength, const IndexSet& indices, bool isRC,
                 size_t coverage, std::vector<std::string>& ovlStr)
{
    std::string initKmer;
    Interval fwdInterval, rvcInterval;
    size_t seedOffSet = 0;

    if(isRC)
        initKmer = reverse_complement(query.substr(query.length() - seedSize - seedOffSet, seedSize));
    else
        initKmer = query.substr(0 + seedOffSet, seedSize);

    fwdInterval = indices.rbwt->find_interval(reverse_str(initKmer));
    rvcInterval = indices.bwt->find_interval(reverse_complement(initKmer));

    // extend each SA index via LF mapping
    for(int64_t fwdRootIndex = fwdInterval.lower;
        fwdInterval.valid() && fwdRootIndex <= fwdInterval.upper && (fwdRootIndex - fwdInterval.lower < (int)coverage);
        fwdRootIndex++) {
        std::string currStr = initKmer;
        currStr.reserve(maxLength);
        int64_t fwdIndex = fwdRootIndex;
        for(size_t currentLength = initKmer.length(); currentLength < maxLength; currentLength++) {
            char b = indices.rbwt->get_char(fwdIndex);
            if(b == '$') break;
            currStr.append(1, b);
            fwdIndex = indices.rbwt->pc(bwt_rank_of(b)) + indices.rbwt->occ(bwt_rank_of(b), fwdIndex - 1);
        }
        if(isRC)
            ovlStr.push_back(reverse_complement(currStr));
        else
            ovlStr.push_back(currStr);
    }

    // LF-mapping of each rvc index
    for(int64_t rvcRootIndex = rvcInterval.lower;
        rvcRootIndex <= rvcInterval.upper && rvcInterval.valid() && (rvcRootIndex - rvcInterval.lower < (int)coverage);
        rvcRootIndex++) {
        std::string currStr = reverse_complement(initKmer);
        currStr.reserve(maxLength);
        int64_t rvcIndex = rvcRootIndex;
        for(size_t currentLength = initKmer.length(); currentLength < maxLength; currentLength++) {
            char b = indices.bwt->get_char(rvcIndex);
            if(b == '$') break;
            currStr = b + currStr;   // in reverse complement, currStr is before b
            rvcIndex = indices.bwt->pc(bwt_rank_of(b)) + indices.bwt->occ(bwt_rank_of(b), rvcIndex - 1);
        }
        if(isRC)
            ovlStr.push_back(currStr);
        else
            ovlStr.push_back(reverse_complement(currStr));
    }
}

void retrieveMatches(const std::string& query, size_t k, size_t min_overlap, double min_identity, size_t coverage,
                     const IndexSet& indices, bool isRC, std::vector<SequenceOverlapPair>& overlap_vector)
{
    std::vector<std::string> ovlStr;
    size_t maxLength = query.length() * 1.1 + 20;
    retrieveStr(query, k, maxLength, indices, isRC, coverage, ovlStr);

    for(std::vector<std::string>::iterator iter = ovlStr.begin(); iter != ovlStr.end(); ++iter) {
        std::string match_sequence = *iter;

        // Ignore identical sequence from forward or backward extension
        if((!isRC && match_sequence.substr(0, query.length()) == query) ||
           (isRC && match_sequence.length() >= query.length() &&
            match_sequence.substr(match_sequence.length() - query.length()) == query))
            continue;

        SequenceOverlap overlap;
        size_t bandwidth = 200;
        // banded global DP alignment, PB requires large mismatch penalty -8
        if(isRC)
            overlap = extend_match(query, match_sequence, query.length() - k, match_sequence.length() - k, bandwidth, 1, -1, -8);
        else
            overlap = extend_match(query, match_sequence, 0, 0, bandwidth, 1, -1, -8);

        bool bPassedOverlap = (size_t)overlap.getOverlapLength() >= (size_t)min_overlap;
        bool bPassedIdentity = overlap.getPercentIdentity() / 100 >= min_identity;

        if(bPassedOverlap && bPassedIdentity) {
            SequenceOverlapPair op;
            op.sequence[1] = match_sequence;
            op.overlap = overlap;
            op.is_reversed = false;
            overlap_vector.push_back(op);
        }
    }
}

MultipleAlignment buildMultipleAlignment(const std::string& query, size_t srcKmerLength, size_t tarKmerLength,
                                         size_t min_overlap, double min_identity, size_t coverage, const IndexSet& indices)
{
    MultipleAlignment multiple_alignment;
    multiple_alignment.addBaseSequence("query", query, "");

    // forward overlap from source seed
    std::vector<SequenceOverlapPair> overlap_vector;
    retrieveMatches(query, srcKmerLength, min_overlap, min_identity, coverage, indices, false, overlap_vector);
    size_t srcSize = overlap_vector.size();

    // reverse overlap from target seed
    retrieveMatches(query, tarKmerLength, min_overlap, min_identity, coverage, indices, true, overlap_vector);

    for(size_t i = 0; i < srcSize; ++i)
        multiple_alignment.addOverlap("Src", overlap_vector[i].sequence[1], "", overlap_vector[i].overlap);
    for(size_t i = srcSize; i < overlap_vector.size(); ++i)
        multiple_alignment.addOverlap("Tar", overlap_vector[i].sequence[1], "", overlap_vector[i].overlap);
    return multiple_alignment;
}

} // namespace lrsc_oracle
