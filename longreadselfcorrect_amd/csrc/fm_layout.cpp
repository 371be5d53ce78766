// fm_layout.cpp -- builds the HBM rank-block image from the reference's on-disk RL-BWT.
//
// Replaces RLBWT::initializeFMIndex (SuffixTools/RLBWT.cpp:109-248): instead of placing
// small/large markers beside the run string, the runs are decoded once and re-packed into
// self-contained 64-byte rank blocks.
#include "fm_layout.h"

#include <cstdio>
#include <cstring>
#include <fstream>

#include "../../include/lrsc.h"

namespace lrsc {

int read_bwt_file(const std::string& path, std::vector<uint8_t>& units, uint64_t& num_strings,
                  uint64_t& num_symbols, std::string& err)
{
    std::FILE* f = std::fopen(path.c_str(), "rb");
    if(!f) { err = "cannot open " + path; return LRSC_ERR_IO; }
    uint8_t hdr[30];
    if(std::fread(hdr, 1, 30, f) != 30) { std::fclose(f); err = "short BWT header: " + path; return LRSC_ERR_FORMAT; }
    uint16_t magic; uint64_t nruns; int32_t flag;
    std::memcpy(&magic, hdr, 2);
    std::memcpy(&num_strings, hdr + 2, 8);
    std::memcpy(&num_symbols, hdr + 10, 8);
    std::memcpy(&nruns, hdr + 18, 8);
    std::memcpy(&flag, hdr + 26, 4);
    if(magic != 0xCACA) {   // RLBWT_FILE_MAGIC, BWTReader.h:33
        std::fclose(f);
        err = "BWT file is not properly formatted, aborting";   // BWTReaderBinary.cpp:63
        return LRSC_ERR_FORMAT;
    }
    if(nruns == 0 || num_symbols == 0) { std::fclose(f); err = "empty BWT: " + path; return LRSC_ERR_FORMAT; }
    units.resize(nruns);
    const size_t got = std::fread(units.data(), 1, nruns, f);
    std::fclose(f);
    if(got != nruns) { err = "truncated BWT body: " + path; return LRSC_ERR_FORMAT; }
    return LRSC_OK;
}

static inline void set_symbol(Block32& b, unsigned off, unsigned code)
{
    b.w[Block32::lo_index(off >> 5)] |= (uint32_t)(code & 1u) << (off & 31u);
    b.w[Block32::hi_index(off >> 5)] |= (uint32_t)(code >> 1) << (off & 31u);
}
static inline void set_symbol(Block64& b, unsigned off, unsigned code)
{
    b.lo[off >> 5] |= (uint32_t)(code & 1u) << (off & 31u);
    b.hi[off >> 5] |= (uint32_t)(code >> 1) << (off & 31u);
}

template <class Block>
static int build_image_t(const uint8_t* units, uint64_t n_units, uint64_t num_symbols,
                         StrandImage& out, std::string& err)
{
    using CountT = decltype(Block::cnt[0] + 0);
    constexpr uint64_t kSyms = Block::kSyms;
    const uint64_t n_blocks = num_symbols / kSyms + 1;
    out.n_blocks = n_blocks;
    out.n_symbols = num_symbols;
    out.n_runs = n_units;
    out.blocks.assign(n_blocks * sizeof(Block), 0);
    out.dollars.clear();
    Block* blk = reinterpret_cast<Block*>(out.blocks.data());

    uint64_t counts[4] = {0, 0, 0, 0};
    uint64_t n_dollar = 0;
    uint64_t pos = 0;
    uint64_t cur_block = 0;
    for(int c = 0; c < 4; ++c) blk[0].cnt[c] = 0;

    for(uint64_t u = 0; u < n_units; ++u) {
        const unsigned rank = units[u] >> 5;
        const unsigned len = units[u] & 0x1F;
        if(rank > 4 || len == 0) { err = "corrupt RL unit in BWT"; return LRSC_ERR_FORMAT; }
        if(pos + len > num_symbols) { err = "BWT runs exceed the symbol count in the header"; return LRSC_ERR_FORMAT; }
        for(unsigned i = 0; i < len; ++i, ++pos) {
            const uint64_t b = pos / kSyms;
            const unsigned off = (unsigned)(pos % kSyms);
            if(b != cur_block) {
                cur_block = b;
                for(int c = 0; c < 4; ++c) blk[b].cnt[c] = (CountT)counts[c];
            }
            unsigned code = 0;
            if(rank == 0) {
                out.dollars.push_back(pos);
                ++n_dollar;
                blk[b].cnt[0] |= (CountT)((CountT)1 << (sizeof(CountT) * 8 - 1));
            } else {
                code = rank - 1;
                ++counts[code];
            }
            set_symbol(blk[b], off, code);
        }
    }
    if(pos != num_symbols) { err = "BWT runs do not add up to the symbol count in the header"; return LRSC_ERR_FORMAT; }
    // terminal block for Occ(b, N-1) when N is a multiple of the block size
    if(num_symbols % kSyms == 0) {
        const uint64_t b = n_blocks - 1;
        for(int c = 0; c < 4; ++c) blk[b].cnt[c] = (CountT)counts[c];
    }
    // '$' directory: rows before each group of blocks
    {
        const uint64_t n_groups = (n_blocks >> kDollarDirShift) + 2;
        if(out.dollars.size() >= (1ull << 32)) { err = "more than 2^32 reads"; return LRSC_ERR_UNSUPPORTED; }
        out.dollar_dir.assign(n_groups, 0);
        size_t j = 0;
        for(uint64_t g = 0; g < n_groups; ++g) {
            const uint64_t start = (g << kDollarDirShift) * kSyms;
            while(j < out.dollars.size() && out.dollars[j] < start) ++j;
            out.dollar_dir[g] = (uint32_t)j;
        }
    }
    out.pred[0] = 0;
    out.pred[1] = n_dollar;
    out.pred[2] = out.pred[1] + counts[0];
    out.pred[3] = out.pred[2] + counts[1];
    out.pred[4] = out.pred[3] + counts[2];
    return LRSC_OK;
}

int build_strand_image(const uint8_t* units, uint64_t n_units, uint64_t num_symbols, bool wide,
                       StrandImage& out, std::string& err)
{
    if(!wide) {
        if(num_symbols >= (1ull << 31)) { err = "Block32 layout needs < 2^31 symbols"; return LRSC_ERR_ARG; }
        return build_image_t<Block32>(units, n_units, num_symbols, out, err);
    }
    return build_image_t<Block64>(units, n_units, num_symbols, out, err);
}

} // namespace lrsc
