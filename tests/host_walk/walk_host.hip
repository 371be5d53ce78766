// walk_host.hip -- TEST INFRASTRUCTURE ONLY (never linked into or loaded by the product).
//
// Runs the product's seed-to-seed walk -- longreadselfcorrect_amd/csrc/walk_device.h, the very source wp_extend_kernel and
// walk_extend_kernel compile for gfx950 -- on the CPU, one walk at a time, over the same rank-block image and k-mer tables the
// device uses, so that `pytest -m "not gpu"` can hold it against the CPU oracle without a device: both the general step
// (Walk::run, what extend.hip's kernels do) and the single-leaf fast path with hand-over to the general step (the loop of
// wp_extend_kernel).  The header is compiled with LRSC_WALK_FN = __host__ __device__ (tests/host_walk/Makefile); in the product
// it is __device__ only and nothing there can reach this code.  No HIP runtime call is made.
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../longreadselfcorrect_amd/csrc/fm_layout.h"
#include "../../longreadselfcorrect_amd/csrc/walk_device.h"

using namespace lrsc;

struct HostIndex {
    StrandImage image[2];
    bool wide = false;
    FmIndexDev dev{};
    std::vector<std::vector<uint8_t>> tables;
    std::vector<uint32_t> mtab;
};

template <bool WIDE>
static void build_tables(HostIndex* ix, const int* ks, int n)
{
    using P = typename Lay<WIDE>::pos_t;
    ix->mtab.resize(MaskTabSize<WIDE>::value);
    fill_mask_table_serial<WIDE>(ix->mtab.data());
    const StrandC<P> sF = strand_consts<P>(ix->dev.strand[LRSC_RBWT]);
    const StrandC<P> sR = strand_consts<P>(ix->dev.strand[LRSC_BWT]);
    for(int i = 0; i < n && ix->tables.size() < 5; ++i) {
        const uint32_t k = (uint32_t)ks[i];
        if(k == 0 || k > 12) continue;
        const uint64_t n_codes = 1ull << (2 * k);
        const size_t eb = WIDE ? 32 : 16;
        std::vector<uint8_t> buf(n_codes * eb);
        for(uint64_t code = 0; code < n_codes; ++code) {
            WalkState<P> st = walk_init<P>();
            for(uint32_t t = 0; t < k; ++t) {
                const uint32_t c = (uint32_t)(code >> (2 * (k - 1 - t))) & 3u;
                st = walk_step<WIDE>(sF, sR, c, 1u << 30, st, ix->mtab.data());
            }
            if(WIDE) {
                uint64_t e[4] = {(uint64_t)st.fwd.lo, (uint64_t)st.fwd.hi, (uint64_t)st.rvc.lo, (uint64_t)st.rvc.hi};
                std::memcpy(buf.data() + code * 32, e, 32);
            } else {
                uint32_t e[4] = {(uint32_t)st.fwd.lo, (uint32_t)st.fwd.hi, (uint32_t)st.rvc.lo, (uint32_t)st.rvc.hi};
                std::memcpy(buf.data() + code * 16, e, 16);
            }
        }
        ix->tables.push_back(std::move(buf));
    }
    // tables become visible only now: walk_step above must not consult a half-built one
    int slot = 0;
    for(int i = 0; i < n && slot < (int)ix->tables.size(); ++i) {
        const uint32_t k = (uint32_t)ks[i];
        if(k == 0 || k > 12) continue;
        ix->dev.ktab[slot].entries = ix->tables[slot].data();
        ix->dev.ktab[slot].k = k;
        ++slot;
    }
}

extern "C" void* hw_index_create(const uint8_t* bwt_units, uint64_t n0, const uint8_t* rbwt_units, uint64_t n1, uint64_t num_symbols,
                                 int wide, const int* table_ks, int n_tables)
{
    HostIndex* ix = new HostIndex();
    ix->wide = wide != 0;
    std::string err;
    if(build_strand_image(bwt_units, n0, num_symbols, ix->wide, ix->image[0], err) != 0 ||
       build_strand_image(rbwt_units, n1, num_symbols, ix->wide, ix->image[1], err) != 0) { delete ix; return nullptr; }
    std::memset(&ix->dev, 0, sizeof(ix->dev));
    ix->dev.wide = ix->wide ? 1u : 0u;
    for(int s = 0; s < 2; ++s) {
        FmStrand& fs = ix->dev.strand[s];
        fs.blocks = ix->image[s].blocks.data();
        fs.dollars = ix->image[s].dollars.data();
        fs.dollar_dir = ix->image[s].dollar_dir.data();
        fs.dollar_group_syms = (uint64_t)(ix->wide ? Block64::kSyms : Block32::kSyms) << kDollarDirShift;
        fs.n_dollars = ix->image[s].dollars.size();
        fs.n_symbols = ix->image[s].n_symbols;
        fs.n_blocks = ix->image[s].n_blocks;
        for(int c = 0; c < 5; ++c) fs.pred[c] = ix->image[s].pred[c];
    }
    if(ix->wide) build_tables<true>(ix, table_ks, n_tables); else build_tables<false>(ix, table_ks, n_tables);
    return ix;
}
extern "C" void hw_index_free(void* h) { delete static_cast<HostIndex*>(h); }

struct HwParams {                 // the few lrsc_params fields a walk reads (tests fill it from params_default)
    int32_t idmer_len, min_kmer_len, max_leaves, pb_coverage;
    double error_rate;
};

// mode 0: Walk::run (general step only); 1: the loop of wp_extend_kernel (begin_static, begin_root, fast path with hand-over)
template <bool WIDE>
static int run_walk(HostIndex* ix, const HwParams& p, const uint8_t* codes, uint32_t initk, uint32_t path_len, uint32_t trg_len, int32_t dis,
                    uint32_t max_overlap, uint32_t min_sa, int mode, uint8_t* out, uint32_t out_cap, uint32_t* out_len, uint32_t* steps,
                    uint32_t* fast_steps)
{
    using P = typename Lay<WIDE>::pos_t;
    const uint32_t lq = initk + path_len + trg_len;
    const uint32_t seed = (uint32_t)p.idmer_len, mino = (uint32_t)p.min_kmer_len;
    const double maxLength = (1.2 * (dis + 10)) + (double)(2 * (uint64_t)initk);
    const uint32_t pathw = (uint32_t)(((uint64_t)maxLength + 4 + 15) / 16 + 1);
    const uint32_t n9 = lq - seed + 1, n5 = lq - 5 + 1, nT = trg_len - mino + 1;
    std::vector<SortItem> it9f(n9), it9r(n9);
    std::vector<P> term((size_t)nT * 4);
    std::vector<uint16_t> next9f(n9), next9r(n9), head9(512), head5(1024), next5(n5);
    std::vector<uint8_t> flags5(n5);
    std::vector<Leaf<P>> leaves(32 + kMaxChildren);
    std::vector<double> rings(32 * 100);
    std::vector<WalkResultRec> results(kMaxResults);
    std::vector<uint32_t> paths((size_t)(32 + kMaxResults) * pathw), outw(pathw);
    double freqs[101];
    for(int i = 0; i <= 100; ++i) freqs[i] = 0;
    for(int i = p.min_kmer_len; i <= 100; i++) freqs[i] = pow(1 - p.error_rate, i) * (size_t)p.pb_coverage;

    const FmIndexDev& fm = ix->dev;
    const StrandC<P> sf = strand_consts<P>(fm.strand[LRSC_RBWT]);
    const StrandC<P> sr = strand_consts<P>(fm.strand[LRSC_BWT]);
    uint32_t cr = 0, cb = 0;
    for(uint32_t i = 0; i < lq; ++i)
        prepare_offset<WIDE>(fm, sf, sr, ix->mtab.data(), codes, i, lq, initk + path_len, seed, mino, it9f.data(), it9r.data(), flags5.data(),
                             term.data(), cr, cb);

    Walk<WIDE> W;
    W.sF = sf; W.sR = sr; W.fm = &fm; W.mtab = ix->mtab.data();
    W.q = codes;
    W.Lq = lq; W.initk = initk; W.path_len = path_len; W.trg_len = trg_len; W.dis = dis;
    W.seedSize = seed; W.minOverlap = mino; W.maxOverlap = max_overlap; W.maxLeaves = (uint32_t)p.max_leaves;
    W.min_SA_threshold = min_sa;
    W.PBcoverage = (uint64_t)p.pb_coverage; W.PacBioErrorRate = p.error_rate; W.errorRate = 0.25; W.localK = 100;
    W.freqsOfKmerSize = freqs;
    if(dis > 100) W.maxIndelSize = (uint64_t)(dis * 0.2); else W.maxIndelSize = 20;
    W.maxLength = (uint64_t)((1.2 * (dis + 10)) + (double)(2 * (uint64_t)initk));
    W.minLength = (uint64_t)((0.8 * (dis - 20)) + (double)(2 * (uint64_t)initk));
    W.it9f = it9f.data(); W.it9r = it9r.data();
    W.next9f = next9f.data(); W.next9r = next9r.data();
    W.head9f = head9.data(); W.head9r = head9.data() + 256;
    W.head5 = head5.data(); W.next5 = next5.data(); W.flags5 = flags5.data();
    W.term = term.data();
    W.n_term = trg_len >= mino ? trg_len - mino + 1 : 0;
    W.cur = leaves.data(); W.nxt = W.cur + 32; W.leaf_small = W.cur;
    W.rings = rings.data();
    W.paths = paths.data(); W.pathw = pathw; W.rpaths = W.paths + (uint64_t)32 * pathw;
    W.results = results.data();
    W.n_rank = 0; W.n_blk = 0; W.steps = 0; W.leaf_steps = 0; W.max_front = 1; W.error = 0; W.cyc_setup = 0; W.cyc_loop = 0; W.prof = nullptr;
    W.profile = false;

    uint32_t len = 0, mi = 0, n_fast = 0;
    int code;
    if(mode == 0) code = W.run(&len, outw.data(), &mi);
    else {
        W.begin_static();
        W.begin_root(nullptr);
        Leaf<P> L;
        uint32_t pw = 0;
        bool fast = false;
        while(true) {
            if(!fast && W.can_fast()) { W.enter_fast(L, pw); fast = true; }
            int r = 2;
            if(fast) {
                r = W.step_fast(L, pw);
                if(r != 1) fast = false; else ++n_fast;
            }
            if(r == 2) r = W.step() ? 1 : 0;
            if(r != 1) break;
        }
        code = W.finish(&len, outw.data(), &mi);
    }
    *steps = (uint32_t)W.steps;
    *fast_steps = n_fast;
    *out_len = 0;
    if(code > 0) {
        const uint32_t tail_from = mi + mino;
        const uint32_t tail = trg_len > mino && tail_from <= trg_len ? trg_len - tail_from : 0;
        if(len + tail > out_cap) return -1000;
        for(uint32_t i = 0; i < len; ++i) out[i] = (uint8_t)path_get(outw.data(), i);
        const uint8_t* trg = codes + initk + path_len;
        for(uint32_t i = 0; i < tail; ++i) out[len + i] = trg[tail_from + i];
        *out_len = len + tail;
    }
    return code;
}

// codes: beginning k-mer | raw read segment | target seed, as 0..3
extern "C" int hw_extend_walk(void* h, const HwParams* p, const uint8_t* codes, uint32_t initk, uint32_t path_len, uint32_t trg_len, int32_t dis,
                              uint32_t max_overlap, uint32_t min_sa, int mode, uint8_t* out, uint32_t out_cap, uint32_t* out_len, uint32_t* steps,
                              uint32_t* fast_steps)
{
    HostIndex* ix = static_cast<HostIndex*>(h);
    return ix->wide ? run_walk<true>(ix, *p, codes, initk, path_len, trg_len, dis, max_overlap, min_sa, mode, out, out_cap, out_len, steps, fast_steps)
                    : run_walk<false>(ix, *p, codes, initk, path_len, trg_len, dis, max_overlap, min_sa, mode, out, out_cap, out_len, steps, fast_steps);
}
