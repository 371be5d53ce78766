// fm_layout.h -- host image of the FM-index in the HBM block layout (see fm_device.h).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "fm_device.h"

namespace lrsc {

struct StrandImage {
    std::vector<uint8_t> blocks;      // n_blocks * 64 bytes (Block32 or Block64)
    std::vector<uint64_t> dollars;    // sorted positions of '$' rows
    std::vector<uint32_t> dollar_dir; // '$' rows before every group of 2^kDollarDirShift blocks (+ one terminal entry)
    uint64_t n_blocks = 0;
    uint64_t n_symbols = 0;
    uint64_t n_runs = 0;
    uint64_t pred[5] = {0, 0, 0, 0, 0};
};

// Parse the 30-byte header + RL units of a .bwt/.rbwt file
// (format: SuffixTools/BWTWriterBinary.cpp:28-46,82-93 / BWTReaderBinary.cpp:55-85).
// Returns an lrsc_status.
int read_bwt_file(const std::string& path, std::vector<uint8_t>& units, uint64_t& num_strings,
                  uint64_t& num_symbols, std::string& err);

// Re-encode RL units ((rank<<5)|len, rank in $ACGT = 0..4) into rank blocks.
// wide=false -> Block32 (requires num_symbols < 2^31), wide=true -> Block64.
int build_strand_image(const uint8_t* units, uint64_t n_units, uint64_t num_symbols, bool wide,
                       StrandImage& out, std::string& err);

} // namespace lrsc
