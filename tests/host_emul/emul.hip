// emul.hip -- TEST INFRASTRUCTURE ONLY (never linked into or loaded by the product).
//
// Drives the correction kernels' per-lane state machine (longreadselfcorrect_amd/csrc/walk_sm.h -- the very code
// correct_sm.hip runs on the GPU) one lane at a time on the CPU, over the same rank-block image, k-mer tables and
// workspace layout, so that `pytest -m "not gpu"` can compare it with the CPU oracle without a device.
// Built by tests/host_emul/Makefile with hipcc (host side of the HIP headers; no HIP runtime call is made).
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../longreadselfcorrect_amd/csrc/correct_layout.h"
#include "../../longreadselfcorrect_amd/csrc/fm_layout.h"
#include "../../longreadselfcorrect_amd/csrc/walk_sm.h"

using namespace lrsc;

// debugging aid: per-sweep trace of one read (compare with the device kernel's LRSC_SM_TRACE file)
static uint32_t* g_trace = nullptr;
static uint32_t g_trace_cap = 0, g_trace_pos = 1, g_trace_read = 0;
extern "C" void emul_set_trace(uint32_t* buf, uint32_t cap, uint32_t read) { g_trace = buf; g_trace_cap = cap; g_trace_pos = 1; g_trace_read = read; }
extern "C" uint32_t emul_trace_words() { return g_trace_pos; }

struct EmulIndex {
    StrandImage image[2];
    bool wide = false;
    FmIndexDev dev{};
    std::vector<std::vector<uint8_t>> tables;
    std::vector<uint32_t> mtab;
};

template <bool WIDE>
static void build_tables(EmulIndex* ix, const int* ks, int n)
{
    using P = typename Lay<WIDE>::pos_t;
    ix->mtab.resize(MaskTabSize<WIDE>::value);
    fill_mask_table_serial<WIDE>(ix->mtab.data());
    const StrandC<P> sF = strand_consts<P>(ix->dev.strand[LRSC_RBWT]);
    const StrandC<P> sR = strand_consts<P>(ix->dev.strand[LRSC_BWT]);
    int slot = 0;
    for(int i = 0; i < n && slot < 5; ++i) {
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
        ix->dev.ktab[slot].entries = nullptr;     // set after all tables are built (walk_step must not consult them meanwhile)
        ix->dev.ktab[slot].k = 0;
        ++slot;
    }
    slot = 0;
    for(int i = 0; i < n && slot < 5; ++i) {
        const uint32_t k = (uint32_t)ks[i];
        if(k == 0 || k > 12) continue;
        ix->dev.ktab[slot].entries = ix->tables[slot].data();
        ix->dev.ktab[slot].k = k;
        ++slot;
    }
}

extern "C" void* emul_index_create(const uint8_t* bwt_units, uint64_t n0, const uint8_t* rbwt_units, uint64_t n1, uint64_t num_symbols,
                                   int wide, const int* table_ks, int n_tables)
{
    EmulIndex* ix = new EmulIndex();
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
extern "C" void emul_index_free(void* h) { delete static_cast<EmulIndex*>(h); }

// consensus of one correctByMSAlignment call, answered by the caller (the test binds it to the CPU oracle's DP restatement):
// returns the consensus length (codes 0..3 in cons), *n_rows = MultipleAlignment::getNumRows()
typedef int (*emul_dp_cb)(void* user, const uint8_t* query, uint32_t lq, uint32_t k, uint32_t min_overlap, double min_identity,
                          int32_t min_call_coverage, uint32_t* n_rows, uint8_t* cons, uint32_t cons_cap);

template <bool WIDE>
static int run_emul(EmulIndex* ix, const lrsc_params& p, const uint8_t* codes, const uint64_t* off, uint32_t n, const uint32_t* seed_count,
                    const int32_t* seeds_flat, emul_dp_cb cb, void* user, uint32_t max_walks, uint32_t max_steps, int64_t* counters,
                    uint8_t* out_codes, uint64_t out_cap, uint64_t* piece_off, uint64_t piece_cap, uint64_t* n_pieces_out, uint64_t* stats)
{
    using P = typename Lay<WIDE>::pos_t;
    const size_t psz = sizeof(P), lbytes = sizeof(Leaf<P>);
    // seeds in the kernels' slab layout with min_k = 1
    const uint64_t total = off[n];
    std::vector<int32_t> slab((total + n + 1) * kSeedInts, 0);
    std::vector<ReadWork> work(n);
    uint64_t ws_total = 0, out_total = 0, piece_total = 0, sp = 0;
    for(uint32_t r = 0; r < n; ++r) {
        const int32_t* sr = seeds_flat + sp * kSeedInts;
        std::memcpy(slab.data() + seed_slab(off[r], r, 1) * kSeedInts, sr, (size_t)seed_count[r] * kSeedInts * sizeof(int32_t));
        ReadWork& w = work[r];
        std::memset(&w, 0, sizeof(w));
        w.out_off = out_total; w.piece_off = piece_total; w.ws_off = ws_total;
        if(seed_count[r] >= 2) {
            const ReadPlan plan = plan_read(sr, seed_count[r], p.next_target);
            const char* err = nullptr;
            const size_t o = layout_read_work(w, off[r + 1] - off[r], seed_count[r], plan, p.no_dp != 0, p.split != 0, (uint32_t)p.idmer_len, psz, lbytes, &err);
            if(err) return -1;
            out_total += ((uint64_t)w.out_cap + 15) & ~15ull;
            piece_total += w.piece_cap;
            ws_total += o;
        }
        sp += seed_count[r];
    }
    std::vector<uint8_t> ws(ws_total + 64), oc(out_total + 64);
    std::vector<uint32_t> pieces(piece_total + 1);
    std::vector<ReadOut> ro(n);
    std::memset(ro.data(), 0, n * sizeof(ReadOut));
    double freqs[101];
    for(int i = 0; i <= 100; ++i) freqs[i] = 0;
    for(int i = p.min_kmer_len; i <= 100; i++) freqs[i] = pow(1 - p.error_rate, i) * (size_t)p.pb_coverage;

    CorrectArgs a{};
    a.codes = codes; a.read_off = off; a.seeds = slab.data(); a.seed_count = seed_count; a.order = nullptr; a.work = work.data();
    a.n_reads = n; a.min_k = 1; a.reads_per_wave = 64; a.workspace = ws.data(); a.out_codes = oc.data(); a.piece_start = pieces.data();
    a.out = ro.data();
    a.seed_size = (uint32_t)p.idmer_len; a.min_overlap = (uint32_t)p.min_kmer_len; a.max_leaves = (uint32_t)p.max_leaves;
    a.start_kmer_len = p.start_kmer_len; a.next_target = p.next_target; a.split = p.split; a.no_dp = p.no_dp;
    a.setup_quorum_pct = 40; a.max_walks = max_walks; a.max_steps = max_steps;
    a.pb_coverage = (uint64_t)p.pb_coverage; a.pacbio_error_rate = p.error_rate; a.freqs_of_kmer_size = freqs;

    std::vector<uint32_t> dp_index(n, 0);
    std::vector<DpRequest> reqs;
    std::vector<DpMsaOut> msa;
    std::vector<uint8_t> cons;
    uint64_t sweeps = 0, requests = 0, launches = 0;
    std::vector<uint32_t> todo(n);
    for(uint32_t r = 0; r < n; ++r) todo[r] = r;
    const StrandC<P> sF = strand_consts<P>(ix->dev.strand[LRSC_RBWT]);
    const StrandC<P> sR = strand_consts<P>(ix->dev.strand[LRSC_BWT]);
    while(!todo.empty()) {
        ++launches;
        a.dp_index = dp_index.data(); a.dp_reqs = reqs.data(); a.dp_msa = msa.data(); a.dp_cons = cons.data();
        for(uint32_t r : todo) {
            ReadSM<WIDE> L;
            L.n_rank = L.n_blk = L.n_tab = 0; L.tkp = nullptr;
            P ex[16];
            L.init(&ix->dev, &a, &sF, &sR, r);
            SmReq<P> res{};
            uint64_t guard = 0;
            while(L.pc != PC_DONE) {
                const bool have = L.req.kind != kReqNone;
                if(have) { sm_answer<WIDE>(ix->dev, sF, sR, ix->mtab.data(), L.req, res, ex, 1, L.n_rank, L.n_blk, L.n_tab); ++requests; }
                if(g_trace && launches == 1 && r == g_trace_read) sm_trace<P>(g_trace, g_trace_cap, g_trace_pos, L.pc, have, L.req, res);
                L.sweep(have, res, true, true, true, true, ex, 1);
                ++sweeps;
                if(++guard > (1ull << 34)) return -2;
            }
        }
        a.resume = 1;
        // next round: yielded reads go on, parked reads get their DP answer (capi.cpp's round loop)
        std::vector<uint32_t> nxt;
        reqs.clear(); msa.clear();
        uint64_t cons_total = 0;
        for(uint32_t r : todo) {
            const ReadOut& o = ro[r];
            if(o.error != 0 || o.state == kReadDone) continue;
            nxt.push_back(r);
            if(o.state != kReadParked) continue;
            DpRequest q;
            std::memset(&q, 0, sizeof(q));
            q.lq = o.dp_lq; q.k = o.dp_k;
            q.cons_cap = dp_cons_capacity(q.lq);
            q.cons_off = cons_total;
            cons_total += q.cons_cap;
            dp_index[r] = (uint32_t)reqs.size();
            reqs.push_back(q);
        }
        cons.assign(cons_total + 1, 0);
        msa.resize(reqs.size());
        for(uint32_t r : nxt) {
            const ReadOut& o = ro[r];
            if(o.state != kReadParked) continue;
            const DpRequest& q = reqs[dp_index[r]];
            const size_t tot = (size_t)o.dp_total_freq;
            double identity = 0.65;
            size_t min_call_coverage = 15;
            identity += (tot > 50 ? 0.05 : 0);
            identity += (tot > 100 ? 0.05 : 0);
            min_call_coverage = tot > 50 ? tot * 0.4 : min_call_coverage;
            if(!cb) return -3;
            uint32_t rows = 0;
            const int len = cb(user, ws.data() + work[r].ws_off + work[r].o_dpq, q.lq, q.k, q.lq / 10, identity, (int32_t)min_call_coverage, &rows,
                               cons.data() + q.cons_off, q.cons_cap);
            DpMsaOut& m = msa[dp_index[r]];
            std::memset(&m, 0, sizeof(m));
            if(len < 0) m.error = 1; else { m.n_rows = rows; m.cons_len = (uint32_t)len; }
        }
        todo.swap(nxt);
    }
    // results
    uint64_t used = 0, n_pieces = 0;
    for(uint32_t r = 0; r < n; ++r) {
        const ReadOut& o = ro[r];
        if(o.error != 0) return -100 + (o.error < -110 || o.error > -100 ? -9 : (o.error + 100));
        for(int j = 0; j < 10; ++j) counters[(uint64_t)r * 11 + j] = o.c[j];
        counters[(uint64_t)r * 11 + 10] = o.merge;
        for(uint32_t j = 0; j < o.n_pieces; ++j) {
            if(n_pieces >= piece_cap) return -4;
            piece_off[n_pieces++] = used + pieces[work[r].piece_off + j];
        }
        if(used + o.out_len > out_cap) return -4;
        std::memcpy(out_codes + used, oc.data() + work[r].out_off, o.out_len);
        used += o.out_len;
        // piece count of this read is recoverable from merge + the offsets: store it after the counters? (kept simple: one
        // piece per merged read unless --split, where the caller walks piece_off with n_pieces_of)
    }
    if(n_pieces >= piece_cap) return -4;
    piece_off[n_pieces] = used;
    *n_pieces_out = n_pieces;
    if(stats) { stats[0] = sweeps; stats[1] = requests; stats[2] = launches; }
    return 0;
}

extern "C" int emul_correct_reads(void* h, const lrsc_params* p, const uint8_t* codes, const uint64_t* off, uint32_t n, const uint32_t* seed_count,
                                  const int32_t* seeds_flat, emul_dp_cb cb, void* user, uint32_t max_walks, uint32_t max_steps, int64_t* counters,
                                  uint32_t* pieces_per_read, uint8_t* out_codes, uint64_t out_cap, uint64_t* piece_off, uint64_t piece_cap,
                                  uint64_t* n_pieces_out, uint64_t* stats)
{
    EmulIndex* ix = static_cast<EmulIndex*>(h);
    (void)pieces_per_read;
    return ix->wide ? run_emul<true>(ix, *p, codes, off, n, seed_count, seeds_flat, cb, user, max_walks, max_steps, counters, out_codes, out_cap, piece_off, piece_cap, n_pieces_out, stats)
                    : run_emul<false>(ix, *p, codes, off, n, seed_count, seeds_flat, cb, user, max_walks, max_steps, counters, out_codes, out_cap, piece_off, piece_cap, n_pieces_out, stats);
}
