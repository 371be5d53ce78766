// capi.cpp -- implementation of include/lrsc.h on top of the gfx950 kernels.
//
// There is deliberately NO CPU fallback in this file: every compute entry point needs a
// HIP device and returns LRSC_ERR_DEVICE (with the HIP error text in lrsc_last_error())
// when there is none.
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <limits>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/lrsc.h"
#include "fm_layout.h"
#include "kernels.h"
#include "extend.h"
#include "correct_dev.h"
#include "dp_dev.h"
#include "wp.h"
#include "introsort_emul.h"

using namespace lrsc;

static thread_local std::string g_last_error;

static int check_offsets(const uint64_t* off, uint32_t n_reads);

static int fail(int status, const std::string& msg)
{
    g_last_error = msg;
    return status;
}
static int hip_fail(hipError_t e, const char* what)
{
    g_last_error = std::string(what) + ": " + hipGetErrorString(e);
    return LRSC_ERR_DEVICE;
}
#define HIP_TRY(expr)                                              \
    do {                                                           \
        hipError_t _e = (expr);                                    \
        if(_e != hipSuccess) return hip_fail(_e, #expr);           \
    } while(0)

// ---------------------------------------------------------------------------------------
// objects
// ---------------------------------------------------------------------------------------
struct DeviceCopy {
    void* blocks[2] = {nullptr, nullptr};
    uint64_t* dollars[2] = {nullptr, nullptr};
    uint32_t* dollar_dir[2] = {nullptr, nullptr};
    void* ktab[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    FmIndexDev dev{};
};

struct lrsc_index {
    StrandImage image[2];     // [LRSC_BWT], [LRSC_RBWT]
    uint64_t num_strings = 0;
    uint64_t num_symbols = 0;
    bool wide = false;
    std::mutex mu;
    std::map<int, DeviceCopy> copies;
};

template <class T>
struct DevBuf {
    T* p = nullptr;
    size_t cap = 0;
    ~DevBuf() { if(p) (void)hipFree(p); }
    hipError_t reserve(size_t n)
    {
        if(n <= cap) return hipSuccess;
        if(p) { (void)hipFree(p); p = nullptr; cap = 0; }
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&p), n * sizeof(T));
        if(e == hipSuccess) cap = n;
        return e;
    }
};

struct CorrectScratch;                                   // device buffers lrsc_batch_correct keeps between calls (defined with it)
static void free_correct_scratch(CorrectScratch* cs);

struct lrsc_ctx {
    CorrectScratch* cs = nullptr;
    const lrsc_index* index = nullptr;
    lrsc_params params{};
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    FmIndexDev fm{};
    DevCounters* d_ctr = nullptr;
    lrsc_kernel_stats stats[LRSC_K_COUNT]{};
    // reusable device scratch
    DevBuf<uint8_t> s_in, s_codes, s_out;
    DevBuf<uint64_t> s_off;
    DevBuf<uint32_t> s_chunk;
    DevBuf<int> s_flag;
};

struct lrsc_batch {
    lrsc_ctx* ctx = nullptr;
    uint32_t n_reads = 0;
    uint64_t total_bases = 0;
    uint8_t* d_codes = nullptr;
    uint64_t* d_off = nullptr;
    uint32_t* d_chunk = nullptr;
    // compact grid features (resident)
    uint32_t n_k = 0;
    uint8_t ks[kMaxPool]{};
    int8_t freq_index[kMaxPool]{};
    int8_t row_of_k[64]{};
    uint32_t n_rows = 0;
    int32_t* d_freq = nullptr;
    uint8_t* d_base_counted = nullptr;
    uint8_t* d_valid = nullptr;
    bool grid_done = false;
    // seed finding
    uint32_t min_k = 1;
    uint64_t seed_cap = 0;
    unsigned long long* d_flags = nullptr;
    uint32_t* d_zeros = nullptr;
    uint8_t* d_attr = nullptr;
    unsigned long long* d_start_bits = nullptr;
    int32_t* d_seeds = nullptr;
    uint32_t* d_seed_count = nullptr;
    float* d_thr = nullptr;
    void* d_scan_tmp = nullptr;
    size_t scan_tmp_cap = 0;
    bool seeds_done = false;
    // --debugseed collection (lrsc_batch_set_debug)
    int debug_flags = 0;
    int32_t* d_outcasts = nullptr;
    uint32_t* d_outcast_count = nullptr;
    uint8_t* d_walk_log = nullptr;
    float* d_ratio = nullptr;
    bool walk_log_done = false;
};

// ---------------------------------------------------------------------------------------
// misc
// ---------------------------------------------------------------------------------------
extern "C" const char* lrsc_strerror(int status)
{
    switch(status) {
        case LRSC_OK: return "ok";
        case LRSC_ERR_IO: return "I/O error";
        case LRSC_ERR_FORMAT: return "BWT file is not properly formatted";
        case LRSC_ERR_ARG: return "invalid argument";
        case LRSC_ERR_NOMEM: return "out of memory";
        case LRSC_ERR_DEVICE: return "HIP device error";
        case LRSC_ERR_CAPACITY: return "output buffer too small";
        case LRSC_ERR_UNSUPPORTED: return "unsupported";
        case LRSC_ERR_LIMIT: return "internal capacity exceeded";
        default: return "unknown error";
    }
}
extern "C" const char* lrsc_last_error(void) { return g_last_error.c_str(); }
extern "C" int lrsc_abi_version(void) { return LRSC_ABI_VERSION; }

extern "C" int lrsc_params_default(int genome, int coverage, lrsc_params* out)
{
    if(!out) return fail(LRSC_ERR_ARG, "null params");
    int order;
    switch(genome) {                                  // opt::order, PacBioSelfCorrection.cpp:104
        case 5: order = 0; break;
        case 10: order = 1; break;
        case 100: order = 2; break;
        default: return fail(LRSC_ERR_ARG, "genome must be 5, 10 or 100");
    }
    static const int size[3] = {17, 19, 21};         // opt::size, :105
    std::memset(out, 0, sizeof(*out));
    out->pb_coverage = coverage;
    out->error_rate = 0.15;
    out->start_kmer_len = size[order];               // :197
    out->offset[0] = 0;
    out->offset[1] = 2 * std::min(std::max(coverage / 30 - 1, 0), order + 1);   // :198
    out->offset[2] = -2 * (order + 1);               // :199
    out->mode = 1;
    out->manual = 0;
    out->scan_kmer_len = 19;
    out->kmer_len_up_bound = 50;
    out->radius = 100;
    out->hh_ratio = 0.6f;
    out->next_target = 1;
    out->max_leaves = 32;
    out->idmer_len = 9;
    out->min_kmer_len = 13;
    out->split = 0;
    out->no_dp = 0;
    return LRSC_OK;
}

// ---------------------------------------------------------------------------------------
// index
// ---------------------------------------------------------------------------------------
static int index_from_units_impl(const uint8_t* u0, uint64_t n0, const uint8_t* u1, uint64_t n1,
                                 uint64_t num_strings, uint64_t num_symbols, lrsc_index** out)
{
    lrsc_index* idx = new(std::nothrow) lrsc_index();
    if(!idx) return fail(LRSC_ERR_NOMEM, "lrsc_index");
    idx->num_strings = num_strings;
    idx->num_symbols = num_symbols;
    // Block64 (64-bit counters, 128 symbols per block) from 2^31 symbols per strand; LRSC_FORCE_WIDE=1 selects it for any
    // index so that the wide code path can be tested on small data
    idx->wide = num_symbols >= (1ull << 31) || std::getenv("LRSC_FORCE_WIDE") != nullptr;
    int st[2] = {LRSC_OK, LRSC_OK};
    std::string err[2];
    const uint8_t* us[2] = {u0, u1};
    const uint64_t ns[2] = {n0, n1};
    // the two strands are independent: build them on two host threads
    std::thread t([&]() { st[1] = build_strand_image(us[1], ns[1], num_symbols, idx->wide, idx->image[1], err[1]); });
    st[0] = build_strand_image(us[0], ns[0], num_symbols, idx->wide, idx->image[0], err[0]);
    t.join();
    for(int s = 0; s < 2; ++s)
        if(st[s] != LRSC_OK) { const int r = fail(st[s], err[s]); delete idx; return r; }
    if(idx->image[0].dollars.size() != num_strings || idx->image[1].dollars.size() != num_strings) {
        delete idx;
        return fail(LRSC_ERR_FORMAT, "number of '$' rows differs from the number of strings in the header");
    }
    *out = idx;
    return LRSC_OK;
}

extern "C" int lrsc_index_from_units(const uint8_t* bwt_units, uint64_t n_bwt_units, const uint8_t* rbwt_units,
                                     uint64_t n_rbwt_units, uint64_t num_strings, uint64_t num_symbols,
                                     lrsc_index** out)
{
    if(!bwt_units || !rbwt_units || !out || num_symbols == 0) return fail(LRSC_ERR_ARG, "null/empty index input");
    return index_from_units_impl(bwt_units, n_bwt_units, rbwt_units, n_rbwt_units, num_strings, num_symbols, out);
}

extern "C" int lrsc_index_open(const char* bwt_path, const char* rbwt_path, lrsc_index** out)
{
    if(!bwt_path || !rbwt_path || !out) return fail(LRSC_ERR_ARG, "null path");
    std::vector<uint8_t> u[2];
    uint64_t nstr[2] = {0, 0}, nsym[2] = {0, 0};
    std::string err;
    int st = read_bwt_file(bwt_path, u[0], nstr[0], nsym[0], err);
    if(st != LRSC_OK) return fail(st, err);
    st = read_bwt_file(rbwt_path, u[1], nstr[1], nsym[1], err);
    if(st != LRSC_OK) return fail(st, err);
    if(nstr[0] != nstr[1] || nsym[0] != nsym[1]) return fail(LRSC_ERR_FORMAT, ".bwt and .rbwt disagree on strings/symbols");
    return index_from_units_impl(u[0].data(), u[0].size(), u[1].data(), u[1].size(), nstr[0], nsym[0], out);
}

extern "C" int lrsc_index_info_get(const lrsc_index* idx, lrsc_index_info* out)
{
    if(!idx || !out) return fail(LRSC_ERR_ARG, "null");
    std::memset(out, 0, sizeof(*out));
    out->num_strings = idx->num_strings;
    out->num_symbols = idx->num_symbols;
    for(int s = 0; s < 2; ++s) {
        out->num_runs[s] = idx->image[s].n_runs;
        for(int c = 0; c < 5; ++c) out->pred_count[s][c] = idx->image[s].pred[c];
        out->device_bytes += idx->image[s].blocks.size() + idx->image[s].dollars.size() * 8 + idx->image[s].dollar_dir.size() * 4;
    }
    out->block_bytes = 64;
    out->block_symbols = idx->wide ? Block64::kSyms : Block32::kSyms;
    return LRSC_OK;
}

extern "C" int lrsc_index_upload(lrsc_index* idx, int device)
{
    if(!idx) return fail(LRSC_ERR_ARG, "null index");
    std::lock_guard<std::mutex> lock(idx->mu);
    if(idx->copies.count(device)) return LRSC_OK;
    HIP_TRY(hipSetDevice(device));
    DeviceCopy dc;
    dc.dev.wide = idx->wide ? 1u : 0u;
    for(int s = 0; s < 2; ++s) {
        const StrandImage& im = idx->image[s];
        HIP_TRY(hipMalloc(&dc.blocks[s], im.blocks.size()));
        HIP_TRY(hipMemcpy(dc.blocks[s], im.blocks.data(), im.blocks.size(), hipMemcpyHostToDevice));
        const size_t db = std::max<size_t>(im.dollars.size(), 1) * sizeof(uint64_t);
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&dc.dollars[s]), db));
        if(!im.dollars.empty())
            HIP_TRY(hipMemcpy(dc.dollars[s], im.dollars.data(), im.dollars.size() * sizeof(uint64_t), hipMemcpyHostToDevice));
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&dc.dollar_dir[s]), im.dollar_dir.size() * sizeof(uint32_t)));
        HIP_TRY(hipMemcpy(dc.dollar_dir[s], im.dollar_dir.data(), im.dollar_dir.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        FmStrand& fs = dc.dev.strand[s];
        fs.blocks = dc.blocks[s];
        fs.dollars = dc.dollars[s];
        fs.dollar_dir = dc.dollar_dir[s];
        fs.dollar_group_syms = (uint64_t)(idx->wide ? Block64::kSyms : Block32::kSyms) << kDollarDirShift;
        fs.n_dollars = im.dollars.size();
        fs.n_symbols = im.n_symbols;
        fs.n_blocks = im.n_blocks;
        for(int c = 0; c < 5; ++c) fs.pred[c] = im.pred[c];
    }
    // k-mer interval tables (narrow indexes): sizes 5 and 9 (the walk's 5-mer / idmer look-ups), T = floor(log4 N) clamped
    // to [9, 13] -- up to there practically every k-mer occurs in the index -- and T + 2, where most chance matches have
    // died (16 bytes x 4^k: 1.07 GB at 13, 17.2 GB at 15, taken only if it fits a quarter of the free HBM).
    // LRSC_KTAB_K overrides T (0 disables all tables), LRSC_KTAB_K2 the fourth size (0 disables it).
    {
        const size_t entry_bytes = idx->wide ? 32 : 16;          // 4 x u64 for Block64 indexes
        int T = 0;
        for(uint64_t n = idx->num_symbols; n >= 4; n >>= 2) ++T;
        T = std::max(9, std::min(13, T));
        if(const char* e = std::getenv("LRSC_KTAB_K")) T = std::atoi(e);
        int T2 = T > 0 ? std::min(15, T + 2) : 0;
        if(const char* e = std::getenv("LRSC_KTAB_K2")) T2 = std::atoi(e);
        // LRSC_KTAB_K3 (experimental, default off): a fifth table of T + 3 (16-mers: 69 GB)
        int T3 = 0;
        if(const char* e = std::getenv("LRSC_KTAB_K3")) T3 = std::atoi(e);
        uint32_t want[5] = {5, 9, (uint32_t)T, (uint32_t)T2, (uint32_t)T3};
        uint32_t ks[5] = {0, 0, 0, 0, 0};
        uint32_t n_t = 0;
        for(int i = 0; i < 5 && T > 0; ++i) {
            if(want[i] == 0 || want[i] > 16 || (n_t > 0 && want[i] <= ks[n_t - 1])) continue;
            const size_t bytes = entry_bytes << (2 * want[i]);
            if(i >= 3) {
                size_t free_b = 0, total_b = 0;
                HIP_TRY(hipMemGetInfo(&free_b, &total_b));
                if(bytes > free_b / (i == 3 ? 4 : 3)) continue;
            }
            HIP_TRY(hipMalloc(&dc.ktab[n_t], bytes));
            // each table starts from the previous (smaller) one; dc.dev.ktab[].k stays 0 until all are built so that the
            // builder's own walk_step never consults a table
            hipError_t e2 = launch_ktab_build(dc.dev, want[i], dc.ktab[n_t], n_t ? ks[n_t - 1] : 0, n_t ? dc.ktab[n_t - 1] : nullptr, nullptr);
            if(e2 != hipSuccess) return hip_fail(e2, "ktab build");
            dc.dev.ktab[n_t].entries = dc.ktab[n_t];
            dc.dev.ktab[n_t].k = 0;
            ks[n_t] = want[i];
            ++n_t;
        }
        HIP_TRY(hipDeviceSynchronize());
        for(uint32_t i = 0; i < n_t; ++i) dc.dev.ktab[i].k = ks[i];
    }
    idx->copies[device] = dc;
    return LRSC_OK;
}

extern "C" void lrsc_index_close(lrsc_index* idx)
{
    if(!idx) return;
    for(auto& kv : idx->copies) {
        if(hipSetDevice(kv.first) != hipSuccess) continue;
        for(int s = 0; s < 2; ++s) {
            if(kv.second.blocks[s]) (void)hipFree(kv.second.blocks[s]);
            if(kv.second.dollars[s]) (void)hipFree(kv.second.dollars[s]);
            if(kv.second.dollar_dir[s]) (void)hipFree(kv.second.dollar_dir[s]);
        }
        for(int t = 0; t < 5; ++t) if(kv.second.ktab[t]) (void)hipFree(kv.second.ktab[t]);
    }
    delete idx;
}

// ---------------------------------------------------------------------------------------
// index construction
// ---------------------------------------------------------------------------------------
namespace lrsc {
int build_bwt_device(const char* reads, const uint64_t* off, uint32_t n_reads, int reverse_reads, int device,
                     std::vector<uint8_t>& bwt_out, uint32_t* rounds_out, std::string& err);
}

extern "C" int lrsc_build_bwt(const char* reads, const uint64_t* read_off, uint32_t n_reads, int reverse_reads,
                              int device, uint8_t** units_out, uint64_t* n_units_out)
{
    if(!reads || !units_out || !n_units_out || n_reads == 0) return fail(LRSC_ERR_ARG, "null / empty read set");
    int st = check_offsets(read_off, n_reads);
    if(st != LRSC_OK) return st;
    std::vector<uint8_t> bwt;
    std::string err;
    st = build_bwt_device(reads, read_off, n_reads, reverse_reads, device, bwt, nullptr, err);
    if(st != LRSC_OK) return fail(st, err);
    // RL-encode as BWTWriterBinary::writeBWChar does: same symbol and run < 31 extends the run
    uint64_t n_units = 0;
    {
        uint8_t prev = 0xFF; unsigned run = 0;
        for(uint8_t c : bwt) {
            if(c == prev && run < 31) ++run;
            else { ++n_units; prev = c; run = 1; }
        }
    }
    uint8_t* units = static_cast<uint8_t*>(std::malloc(n_units ? n_units : 1));
    if(!units) return fail(LRSC_ERR_NOMEM, "RL units");
    {
        uint64_t u = 0; uint8_t prev = 0xFF; unsigned run = 0;
        for(uint8_t c : bwt) {
            if(c == prev && run < 31) { ++run; units[u - 1] = (uint8_t)((c << 5) | run); }
            else { prev = c; run = 1; units[u++] = (uint8_t)((c << 5) | 1); }
        }
    }
    *units_out = units;
    *n_units_out = n_units;
    return LRSC_OK;
}

extern "C" void lrsc_buffer_free(void* p) { std::free(p); }

extern "C" int lrsc_write_bwt_file(const char* path, const uint8_t* units, uint64_t n_units, uint64_t num_strings,
                                   uint64_t num_symbols)
{
    if(!path || (!units && n_units)) return fail(LRSC_ERR_ARG, "null");
    std::FILE* f = std::fopen(path, "wb");
    if(!f) return fail(LRSC_ERR_IO, std::string("cannot open ") + path);
    uint8_t hdr[30];
    const uint16_t magic = 0xCACA;
    const int32_t flag = 0;   // BWF_NOFMI
    std::memcpy(hdr, &magic, 2);
    std::memcpy(hdr + 2, &num_strings, 8);
    std::memcpy(hdr + 10, &num_symbols, 8);
    std::memcpy(hdr + 18, &n_units, 8);
    std::memcpy(hdr + 26, &flag, 4);
    bool ok = std::fwrite(hdr, 1, 30, f) == 30;
    ok = ok && std::fwrite(units, 1, n_units, f) == n_units;
    ok = (std::fclose(f) == 0) && ok;
    return ok ? LRSC_OK : fail(LRSC_ERR_IO, std::string("short write to ") + path);
}

// ---------------------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------------------
extern "C" int lrsc_ctx_create(const lrsc_index* idx, const lrsc_params* params, int device, lrsc_ctx** out)
{
    if(!idx || !out) return fail(LRSC_ERR_ARG, "null");
    lrsc_index* midx = const_cast<lrsc_index*>(idx);
    FmIndexDev fm;
    {
        std::lock_guard<std::mutex> lock(midx->mu);
        auto it = midx->copies.find(device);
        if(it == midx->copies.end()) return fail(LRSC_ERR_DEVICE, "index not uploaded to this device (call lrsc_index_upload)");
        fm = it->second.dev;
    }
    HIP_TRY(hipSetDevice(device));
    lrsc_ctx* ctx = new(std::nothrow) lrsc_ctx();
    if(!ctx) return fail(LRSC_ERR_NOMEM, "lrsc_ctx");
    ctx->index = idx;
    ctx->device = device;
    ctx->fm = fm;
    if(params) ctx->params = *params;
    else (void)lrsc_params_default(10, 90, &ctx->params);
    hipError_t e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if(e == hipSuccess) e = hipEventCreate(&ctx->ev0);
    if(e == hipSuccess) e = hipEventCreate(&ctx->ev1);
    if(e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&ctx->d_ctr), kCtrShards * sizeof(DevCounters));
    if(e == hipSuccess) e = hipMemset(ctx->d_ctr, 0, kCtrShards * sizeof(DevCounters));
    if(e != hipSuccess) { lrsc_ctx_destroy(ctx); return hip_fail(e, "lrsc_ctx_create"); }
    *out = ctx;
    return LRSC_OK;
}

extern "C" void lrsc_ctx_destroy(lrsc_ctx* ctx)
{
    if(!ctx) return;
    (void)hipSetDevice(ctx->device);
    if(ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    free_correct_scratch(ctx->cs);
    if(ctx->d_ctr) (void)hipFree(ctx->d_ctr);
    if(ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if(ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    if(ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

extern "C" int lrsc_ctx_get_params(const lrsc_ctx* ctx, lrsc_params* out)
{
    if(!ctx || !out) return fail(LRSC_ERR_ARG, "null");
    *out = ctx->params;
    return LRSC_OK;
}

extern "C" int lrsc_ctx_sync(lrsc_ctx* ctx)
{
    if(!ctx) return fail(LRSC_ERR_ARG, "null ctx");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return LRSC_OK;
}

// Bracket one kernel launch with HIP events on the ctx stream and fold the device counters
// into the per-kernel stats.  `launch` enqueues on ctx->stream.
template <class F>
static int timed_launch(lrsc_ctx* ctx, int which, F&& launch)
{
    HIP_TRY(hipMemsetAsync(ctx->d_ctr, 0, kCtrShards * sizeof(DevCounters), ctx->stream));
    HIP_TRY(hipEventRecord(ctx->ev0, ctx->stream));
    hipError_t e = launch();
    if(e != hipSuccess) return hip_fail(e, "kernel launch");
    HIP_TRY(hipEventRecord(ctx->ev1, ctx->stream));
    HIP_TRY(hipEventSynchronize(ctx->ev1));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    std::vector<DevCounters> shards(kCtrShards);
    HIP_TRY(hipMemcpy(shards.data(), ctx->d_ctr, kCtrShards * sizeof(DevCounters), hipMemcpyDeviceToHost));
    DevCounters h{};
    for(const DevCounters& d : shards) { h.rank_queries += d.rank_queries; h.block_loads += d.block_loads; h.table_loads += d.table_loads; }
    lrsc_kernel_stats& s = ctx->stats[which];
    s.launches += 1;
    s.total_ms += ms;
    s.rank_queries += h.rank_queries;
    s.block_loads += h.block_loads;
    s.table_loads += h.table_loads;
    return LRSC_OK;
}

extern "C" int lrsc_ctx_stats(lrsc_ctx* ctx, int kernel, lrsc_kernel_stats* out)
{
    if(!ctx || !out || kernel < 0 || kernel >= LRSC_K_COUNT) return fail(LRSC_ERR_ARG, "bad stats query");
    *out = ctx->stats[kernel];
    return LRSC_OK;
}
extern "C" int lrsc_ctx_stats_reset(lrsc_ctx* ctx)
{
    if(!ctx) return fail(LRSC_ERR_ARG, "null ctx");
    for(auto& s : ctx->stats) s = lrsc_kernel_stats{};
    return LRSC_OK;
}

// ---------------------------------------------------------------------------------------
// FM primitives
// ---------------------------------------------------------------------------------------
extern "C" int lrsc_rank(lrsc_ctx* ctx, const lrsc_rank_query* q, uint64_t n, uint64_t* out)
{
    if(!ctx || (!q && n) || (!out && n)) return fail(LRSC_ERR_ARG, "null");
    if(n == 0) return LRSC_OK;
    const uint64_t N = ctx->index->num_symbols;
    for(uint64_t i = 0; i < n; ++i) {
        const uint8_t b = q[i].base;
        if(q[i].idx < -1 || q[i].idx >= (int64_t)N || (b != 'A' && b != 'C' && b != 'G' && b != 'T') || q[i].strand > 1)
            return fail(LRSC_ERR_ARG, "rank query out of range");
    }
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(ctx->s_in.reserve(n * sizeof(lrsc_rank_query)));
    HIP_TRY(ctx->s_out.reserve(n * sizeof(uint64_t)));
    HIP_TRY(hipMemcpyAsync(ctx->s_in.p, q, n * sizeof(lrsc_rank_query), hipMemcpyHostToDevice, ctx->stream));
    const int st = timed_launch(ctx, LRSC_K_RANK, [&]() {
        return launch_rank(ctx->fm, reinterpret_cast<const lrsc_rank_query*>(ctx->s_in.p), n,
                           reinterpret_cast<uint64_t*>(ctx->s_out.p), ctx->d_ctr, ctx->stream);
    });
    if(st != LRSC_OK) return st;
    HIP_TRY(hipMemcpy(out, ctx->s_out.p, n * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return LRSC_OK;
}

extern "C" int lrsc_bwt_chars(lrsc_ctx* ctx, int strand, const uint64_t* idx, uint64_t n, char* out)
{
    if(!ctx || (!idx && n) || (!out && n) || strand < 0 || strand > 1) return fail(LRSC_ERR_ARG, "null");
    if(n == 0) return LRSC_OK;
    const uint64_t N = ctx->index->num_symbols;
    for(uint64_t i = 0; i < n; ++i)
        if(idx[i] >= N) return fail(LRSC_ERR_ARG, "BWT position out of range");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(ctx->s_in.reserve(n * sizeof(uint64_t)));
    HIP_TRY(ctx->s_out.reserve(n));
    HIP_TRY(hipMemcpyAsync(ctx->s_in.p, idx, n * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
    hipError_t e = launch_bwt_chars(ctx->fm, strand, reinterpret_cast<const uint64_t*>(ctx->s_in.p), n,
                                    reinterpret_cast<char*>(ctx->s_out.p), ctx->stream);
    if(e != hipSuccess) return hip_fail(e, "bwt_chars");
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    HIP_TRY(hipMemcpy(out, ctx->s_out.p, n, hipMemcpyDeviceToHost));
    return LRSC_OK;
}

extern "C" int lrsc_lf_walk(lrsc_ctx* ctx, const uint64_t* rows, const uint8_t* strand, const uint32_t* max_steps,
                            const uint64_t* out_off, uint64_t n, char* out, uint64_t out_cap, uint32_t* out_len)
{
    if(!ctx || ((!rows || !strand || !max_steps || !out_off || !out || !out_len) && n)) return fail(LRSC_ERR_ARG, "null");
    if(n == 0) return LRSC_OK;
    const uint64_t N = ctx->index->num_symbols;
    std::vector<LfJob> jobs(n);
    uint64_t need = 0;
    for(uint64_t i = 0; i < n; ++i) {
        if(rows[i] >= N || strand[i] > 1) return fail(LRSC_ERR_ARG, "LF job out of range");
        jobs[i].row = rows[i]; jobs[i].out_off = out_off[i]; jobs[i].max_steps = max_steps[i]; jobs[i].strand = strand[i];
        need = std::max(need, out_off[i] + max_steps[i]);
    }
    if(need > out_cap) return fail(LRSC_ERR_CAPACITY, "LF output buffer too small");
    HIP_TRY(hipSetDevice(ctx->device));
    DevBuf<LfJob> d_jobs;
    DevBuf<uint8_t> d_out;
    DevBuf<uint32_t> d_len;
    HIP_TRY(d_jobs.reserve(n));
    HIP_TRY(d_out.reserve(std::max<uint64_t>(need, 1)));
    HIP_TRY(d_len.reserve(n));
    HIP_TRY(hipMemcpyAsync(d_jobs.p, jobs.data(), n * sizeof(LfJob), hipMemcpyHostToDevice, ctx->stream));
    const int st = timed_launch(ctx, LRSC_K_LF, [&]() { return launch_lf_walk(ctx->fm, d_jobs.p, n, d_out.p, d_len.p, ctx->d_ctr, ctx->stream); });
    if(st != LRSC_OK) return st;
    HIP_TRY(hipMemcpy(out_len, d_len.p, n * sizeof(uint32_t), hipMemcpyDeviceToHost));
    std::vector<uint8_t> codes(need);
    HIP_TRY(hipMemcpy(codes.data(), d_out.p, need, hipMemcpyDeviceToHost));
    for(uint64_t i = 0; i < n; ++i)
        for(uint32_t t = 0; t < out_len[i]; ++t) out[out_off[i] + t] = "ACGT"[codes[out_off[i] + t] & 3];
    return LRSC_OK;
}

// upload ASCII bases, encode to 2-bit codes on the device, reject non-ACGT
static int upload_and_encode(lrsc_ctx* ctx, const char* ascii, uint64_t n, uint8_t* d_codes)
{
    HIP_TRY(ctx->s_in.reserve(n));
    HIP_TRY(ctx->s_flag.reserve(1));
    HIP_TRY(hipMemsetAsync(ctx->s_flag.p, 0, sizeof(int), ctx->stream));
    HIP_TRY(hipMemcpyAsync(ctx->s_in.p, ascii, n, hipMemcpyHostToDevice, ctx->stream));
    hipError_t e = launch_encode(reinterpret_cast<const char*>(ctx->s_in.p), d_codes, n, ctx->s_flag.p, ctx->stream);
    if(e != hipSuccess) return hip_fail(e, "encode");
    int bad = 0;
    HIP_TRY(hipMemcpyAsync(&bad, ctx->s_flag.p, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    // SeqReader exits on a non-ACGT base (Util/SeqReader.cpp:115-126); the library reports it
    if(bad) return fail(LRSC_ERR_ARG, "sequence contains a base other than A,C,G,T");
    return LRSC_OK;
}

extern "C" int lrsc_find_kmers(lrsc_ctx* ctx, const char* kmers, uint32_t k, uint64_t n, lrsc_biinterval* out)
{
    if(!ctx || (!kmers && n) || (!out && n) || k == 0) return fail(LRSC_ERR_ARG, "null / k == 0");
    if(n == 0) return LRSC_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(ctx->s_codes.reserve(n * k));
    HIP_TRY(ctx->s_out.reserve(n * sizeof(lrsc_biinterval)));
    int st = upload_and_encode(ctx, kmers, n * k, ctx->s_codes.p);
    if(st != LRSC_OK) return st;
    st = timed_launch(ctx, LRSC_K_FIND, [&]() {
        return launch_find_kmers(ctx->fm, ctx->s_codes.p, k, n, reinterpret_cast<lrsc_biinterval*>(ctx->s_out.p),
                                 ctx->d_ctr, ctx->stream);
    });
    if(st != LRSC_OK) return st;
    HIP_TRY(hipMemcpy(out, ctx->s_out.p, n * sizeof(lrsc_biinterval), hipMemcpyDeviceToHost));
    return LRSC_OK;
}

static int check_pool(const uint8_t* ks, uint32_t n_k)
{
    if(!ks || n_k == 0 || n_k > kMaxPool) return fail(LRSC_ERR_ARG, "pool must hold 1..8 k-mer sizes");
    for(uint32_t i = 0; i < n_k; ++i) {
        if(ks[i] == 0 || (i && ks[i] <= ks[i - 1])) return fail(LRSC_ERR_ARG, "pool sizes must be ascending and > 0");
    }
    return LRSC_OK;
}

static int check_offsets(const uint64_t* off, uint32_t n_reads)
{
    if(!off) return fail(LRSC_ERR_ARG, "null read offsets");
    if(off[0] != 0) return fail(LRSC_ERR_ARG, "read_off[0] must be 0");
    for(uint32_t i = 0; i < n_reads; ++i)
        if(off[i + 1] < off[i]) return fail(LRSC_ERR_ARG, "read offsets must be non-decreasing");
    return LRSC_OK;
}

extern "C" int lrsc_kmer_grid(lrsc_ctx* ctx, const char* reads, const uint64_t* read_off, uint32_t n_reads,
                              const uint8_t* ks, uint32_t n_k, lrsc_biinterval* out_iv, uint8_t* out_size,
                              uint8_t* out_count)
{
    if(!ctx) return fail(LRSC_ERR_ARG, "null ctx");
    int st = check_pool(ks, n_k);
    if(st != LRSC_OK) return st;
    if(n_reads == 0) return LRSC_OK;
    st = check_offsets(read_off, n_reads);
    if(st != LRSC_OK) return st;
    const uint64_t total = read_off[n_reads];
    if(total == 0) return LRSC_OK;
    if(!reads) return fail(LRSC_ERR_ARG, "null reads");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(ctx->s_codes.reserve(total));
    HIP_TRY(ctx->s_off.reserve(n_reads + 1));
    const uint64_t n_chunks = (total + (1ull << kChunkShift) - 1) >> kChunkShift;
    HIP_TRY(ctx->s_chunk.reserve(n_chunks));
    st = upload_and_encode(ctx, reads, total, ctx->s_codes.p);
    if(st != LRSC_OK) return st;
    HIP_TRY(hipMemcpyAsync(ctx->s_off.p, read_off, (n_reads + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
    hipError_t e = launch_chunk_table(ctx->s_off.p, n_reads, total, ctx->s_chunk.p, ctx->stream);
    if(e != hipSuccess) return hip_fail(e, "chunk_table");

    const uint64_t recs = total * n_k;
    DevBuf<lrsc_biinterval> d_iv;
    DevBuf<uint8_t> d_size, d_count;
    if(out_iv) HIP_TRY(d_iv.reserve(recs));
    if(out_size) HIP_TRY(d_size.reserve(recs));
    if(out_count) HIP_TRY(d_count.reserve(recs * 4));

    GridArgs a{};
    a.codes = ctx->s_codes.p;
    a.read_off = ctx->s_off.p;
    a.chunk_read = ctx->s_chunk.p;
    a.total_bases = total;
    a.n_reads = n_reads;
    a.n_k = n_k;
    for(uint32_t i = 0; i < n_k; ++i) a.ks[i] = ks[i];
    a.out_iv = d_iv.p;
    a.out_size = d_size.p;
    a.out_count = d_count.p;
    st = timed_launch(ctx, LRSC_K_GRID, [&]() { return launch_kmer_grid(ctx->fm, a, ctx->d_ctr, ctx->stream); });
    if(st != LRSC_OK) return st;
    if(out_iv) HIP_TRY(hipMemcpy(out_iv, d_iv.p, recs * sizeof(lrsc_biinterval), hipMemcpyDeviceToHost));
    if(out_size) HIP_TRY(hipMemcpy(out_size, d_size.p, recs, hipMemcpyDeviceToHost));
    if(out_count) HIP_TRY(hipMemcpy(out_count, d_count.p, recs * 4, hipMemcpyDeviceToHost));
    return LRSC_OK;
}

// ---------------------------------------------------------------------------------------
// resident batches
// ---------------------------------------------------------------------------------------
extern "C" void lrsc_batch_destroy(lrsc_batch* b)
{
    if(!b) return;
    if(b->ctx) (void)hipSetDevice(b->ctx->device);
    if(b->d_codes) (void)hipFree(b->d_codes);
    if(b->d_off) (void)hipFree(b->d_off);
    if(b->d_chunk) (void)hipFree(b->d_chunk);
    void* ptrs[] = {b->d_freq, b->d_base_counted, b->d_valid, b->d_flags, b->d_zeros, b->d_attr, b->d_start_bits, b->d_seeds,
                    b->d_seed_count, b->d_thr, b->d_scan_tmp, b->d_outcasts, b->d_outcast_count, b->d_walk_log, b->d_ratio};
    for(void* q : ptrs) if(q) (void)hipFree(q);
    delete b;
}

extern "C" int lrsc_batch_create(lrsc_ctx* ctx, const char* reads, const uint64_t* read_off, uint32_t n_reads,
                                 lrsc_batch** out)
{
    if(!ctx || !out || n_reads == 0) return fail(LRSC_ERR_ARG, "null / empty batch");
    int st = check_offsets(read_off, n_reads);
    if(st != LRSC_OK) return st;
    const uint64_t total = read_off[n_reads];
    if(total == 0 || !reads) return fail(LRSC_ERR_ARG, "empty batch");
    HIP_TRY(hipSetDevice(ctx->device));
    lrsc_batch* b = new(std::nothrow) lrsc_batch();
    if(!b) return fail(LRSC_ERR_NOMEM, "lrsc_batch");
    b->ctx = ctx;
    b->n_reads = n_reads;
    b->total_bases = total;
    const uint64_t n_chunks = (total + (1ull << kChunkShift) - 1) >> kChunkShift;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&b->d_codes), total);
    if(e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&b->d_off), (n_reads + 1) * sizeof(uint64_t));
    if(e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&b->d_chunk), n_chunks * sizeof(uint32_t));
    if(e != hipSuccess) { lrsc_batch_destroy(b); return hip_fail(e, "batch alloc"); }
    // stream the bases through the ctx staging buffer in slices so host->device staging stays bounded
    const uint64_t slice = 256ull << 20;
    for(uint64_t o = 0; o < total; o += slice) {
        const uint64_t n = std::min(slice, total - o);
        st = upload_and_encode(ctx, reads + o, n, b->d_codes + o);
        if(st != LRSC_OK) { lrsc_batch_destroy(b); return st; }
    }
    e = hipMemcpyAsync(b->d_off, read_off, (n_reads + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream);
    if(e == hipSuccess) e = launch_chunk_table(b->d_off, n_reads, total, b->d_chunk, ctx->stream);
    if(e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if(e != hipSuccess) { lrsc_batch_destroy(b); return hip_fail(e, "batch upload"); }
    *out = b;
    return LRSC_OK;
}

// pool = {5, 9, scan} U {k + offset[0..2]}  (StriDe/PacBioSelfCorrection.cpp:108,204-206)
static uint32_t pool_from_params(const lrsc_params& p, uint8_t* ks)
{
    std::vector<int> v = {5, 9, p.scan_kmer_len};
    for(int i = 0; i < 3; ++i) v.push_back(p.start_kmer_len + p.offset[i]);
    std::sort(v.begin(), v.end());
    v.erase(std::unique(v.begin(), v.end()), v.end());
    uint32_t n = 0;
    for(int x : v) if(x > 0 && x < 256 && n < kMaxPool) ks[n++] = (uint8_t)x;
    return n;
}

// KmerThreshold::initialize(-1, 50, cov) + calculate (PacBio/KmerThreshold.cpp:11-25,43-63,74-79):
// float arithmetic left to right, floored at 2.0f, running minimum over k; entries below k = 15 and
// k = 51 stay 0.0f.  The host computes it once; the kernels only read the table.
static const float kThresholdFormula[3][6] = {
    {0.0004799107143, -0.008037815126, 0.03673552754, 0.1850695903, -1.572552521, 18.0522088},
    {0.0003348214286, -0.009112394958, 0.04286714686, 0.240519958, -1.8793367350, 21.29319228},
    {0.01714285714, -0.6193907563, 2.266956783, 17.28450630, -100.6983493, 1103.571729}};

// KmerThreshold::initialize(s, e, cov, dir) (PacBio/KmerThreshold.cpp:43-63): table[mode][0 .. end+1], zero below max(s, 15) and at
// end + 1, the running minimum ("cavity") of max(formula, 2) in between.
extern "C" int lrsc_kmer_thresholds_range(int coverage, int end, float* out)
{
    if(!out || end < 15) return fail(LRSC_ERR_ARG, "null table / end < 15");
    const int start = 15, stride = end + 2;
    for(int mode = 0; mode < 3; ++mode) {
        for(int k = 0; k < stride; ++k) out[mode * stride + k] = 0.0f;
        float cavity = std::numeric_limits<float>::max();
        const float* f = kThresholdFormula[mode];
        const int x = coverage;
        for(int y = start; y <= end; ++y) {
            float v = f[0] * x * x + f[1] * x * y + f[2] * y * y + f[3] * x + f[4] * y + f[5];
            v = std::fmax(v, 2.0f);
            cavity = std::fmin(cavity, v);
            out[mode * stride + y] = cavity;
        }
    }
    return LRSC_OK;
}

extern "C" int lrsc_kmer_thresholds(int coverage, float* out)
{
    return lrsc_kmer_thresholds_range(coverage, 50, out);         // pbcorrect: initialize(startKmerLen, kmerLenUpBound = 50, ...)
}

static int batch_setup_rows(lrsc_ctx* ctx, lrsc_batch* b)
{
    if(b->n_k != 0) return LRSC_OK;
    const lrsc_params& p = ctx->params;
    b->n_k = pool_from_params(p, b->ks);
    for(auto& x : b->row_of_k) x = -1;
    for(auto& x : b->freq_index) x = -1;
    // rows kept resident: the scan k-mer and the three static k-mer sizes (LongReadProbe.cpp:49,61,141)
    int wanted[4] = {p.scan_kmer_len, p.start_kmer_len + p.offset[0], p.start_kmer_len + p.offset[1],
                     p.start_kmer_len + p.offset[2]};
    int min_k = 1 << 30;
    for(int i = 0; i < 4; ++i) {
        const int k = wanted[i];
        if(k <= 0 || k > 51) return fail(LRSC_ERR_ARG, "k-mer sizes must lie in 1..51");
        if(i > 0 && k < min_k) min_k = k;
        if(b->row_of_k[k] >= 0) continue;
        for(uint32_t j = 0; j < b->n_k; ++j)
            if(b->ks[j] == k) { b->freq_index[j] = (int8_t)b->n_rows; b->row_of_k[k] = (int8_t)b->n_rows; ++b->n_rows; }
        if(b->row_of_k[k] < 0) return fail(LRSC_ERR_ARG, "k-mer size missing from the pool");
    }
    b->min_k = (uint32_t)min_k;
    if(b->total_bases >= (1ull << 32)) return fail(LRSC_ERR_UNSUPPORTED, "a resident batch holds < 2^32 bases");
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&b->d_freq), (size_t)b->n_rows * b->total_bases * sizeof(int32_t)));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&b->d_base_counted), b->total_bases));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&b->d_valid), b->total_bases));
    return LRSC_OK;
}

static GridArgs batch_grid_args(const lrsc_batch* b)
{
    GridArgs a{};
    a.codes = b->d_codes;
    a.read_off = b->d_off;
    a.chunk_read = b->d_chunk;
    a.total_bases = b->total_bases;
    a.n_reads = b->n_reads;
    a.n_k = b->n_k;
    for(uint32_t i = 0; i < b->n_k; ++i) { a.ks[i] = b->ks[i]; a.freq_index[i] = b->freq_index[i]; }
    a.freq = b->d_freq;
    a.base_counted = b->d_base_counted;
    a.valid_mask = b->d_valid;
    return a;
}

extern "C" int lrsc_batch_kmer_grid(lrsc_ctx* ctx, lrsc_batch* b)
{
    if(!ctx || !b || b->ctx != ctx) return fail(LRSC_ERR_ARG, "batch does not belong to this ctx");
    HIP_TRY(hipSetDevice(ctx->device));
    int st = batch_setup_rows(ctx, b);
    if(st != LRSC_OK) return st;
    const GridArgs a = batch_grid_args(b);
    st = timed_launch(ctx, LRSC_K_GRID, [&]() { return launch_kmer_grid(ctx->fm, a, ctx->d_ctr, ctx->stream); });
    if(st == LRSC_OK) b->grid_done = true;
    return st;
}

static int batch_setup_seeds(lrsc_ctx* ctx, lrsc_batch* b)
{
    if(b->d_seeds) return LRSC_OK;
    b->seed_cap = b->total_bases / b->min_k + b->n_reads + 1;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&b->d_flags), b->total_bases * sizeof(unsigned long long)));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&b->d_zeros), b->total_bases * sizeof(uint32_t)));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&b->d_attr), b->total_bases));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&b->d_start_bits), ((b->total_bases + 255) / 256) * 4 * sizeof(unsigned long long)));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&b->d_seeds), b->seed_cap * kSeedInts * sizeof(int32_t)));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&b->d_seed_count), (size_t)b->n_reads * sizeof(uint32_t)));
    if(b->debug_flags & LRSC_DEBUG_RATIO) HIP_TRY(hipMalloc(reinterpret_cast<void**>(&b->d_ratio), b->total_bases * sizeof(float)));
    if(b->debug_flags & LRSC_DEBUG_OUTCASTS) {
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&b->d_outcasts), b->seed_cap * kSeedInts * sizeof(int32_t)));
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&b->d_outcast_count), (size_t)b->n_reads * sizeof(uint32_t)));
    }
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&b->d_thr), 3 * 52 * sizeof(float)));
    float thr[3 * 52];
    (void)lrsc_kmer_thresholds(ctx->params.pb_coverage, thr);
    HIP_TRY(hipMemcpy(b->d_thr, thr, sizeof(thr), hipMemcpyHostToDevice));
    return LRSC_OK;
}

static SeedArgs batch_seed_args(const lrsc_ctx* ctx, const lrsc_batch* b)
{
    const lrsc_params& p = ctx->params;
    SeedArgs a{};
    a.codes = b->d_codes; a.read_off = b->d_off; a.chunk_read = b->d_chunk;
    a.total_bases = b->total_bases; a.n_reads = b->n_reads;
    a.freq = b->d_freq; a.valid_mask = b->d_valid; a.base_counted = b->d_base_counted;
    for(int i = 0; i < 64; ++i) a.row_of_k[i] = b->row_of_k[i];
    a.base_k = b->ks[0];
    a.start_kmer_len = p.start_kmer_len; a.scan_kmer_len = p.scan_kmer_len; a.kmer_len_up_bound = p.kmer_len_up_bound;
    a.pb_coverage = p.pb_coverage; a.mode = p.mode; a.manual = p.manual; a.radius = p.radius;
    for(int i = 0; i < 3; ++i) a.offset[i] = p.offset[i];
    a.hh_ratio = p.hh_ratio;
    a.thresholds = b->d_thr;
    a.flags = b->d_flags; a.zeros = b->d_zeros; a.attribute = b->d_attr; a.start_bits = b->d_start_bits; a.seeds = b->d_seeds; a.seed_count = b->d_seed_count;
    a.outcasts = b->d_outcasts; a.outcast_count = b->d_outcast_count; a.ratio = b->d_ratio;
    return a;
}

// LongReadProbe::searchSeedsWithHybridKmers for every read of the resident batch: grid -> scan-k-mer
// classes -> prefix sums -> per-position attribute -> per-read greedy scan.
extern "C" int lrsc_batch_find_seeds(lrsc_ctx* ctx, lrsc_batch* b)
{
    if(!ctx || !b || b->ctx != ctx) return fail(LRSC_ERR_ARG, "batch does not belong to this ctx");
    if(ctx->params.kmer_len_up_bound > 50) return fail(LRSC_ERR_ARG, "kmer_len_up_bound must be <= 50 (threshold table)");
    if(ctx->params.manual && (ctx->params.mode < 0 || ctx->params.mode > 2)) return fail(LRSC_ERR_ARG, "mode must be 0, 1 or 2");
    int st = lrsc_batch_kmer_grid(ctx, b);
    if(st != LRSC_OK) return st;
    st = batch_setup_seeds(ctx, b);
    if(st != LRSC_OK) return st;
    const SeedArgs a = batch_seed_args(ctx, b);
    st = timed_launch(ctx, LRSC_K_SEEDS, [&]() {
        hipError_t e = launch_seed_modes(a, ctx->stream);
        if(e == hipSuccess) e = scan_seed_flags(b->d_flags, b->d_zeros, b->total_bases, &b->d_scan_tmp, &b->scan_tmp_cap, ctx->stream);
        if(e == hipSuccess) e = launch_seed_attribute(a, ctx->stream);
        if(e == hipSuccess) e = launch_seed_scan(ctx->fm, a, b->min_k, ctx->d_ctr, ctx->stream);
        return e;
    });
    if(st == LRSC_OK) b->seeds_done = true;
    return st;
}

extern "C" int lrsc_batch_seeds(lrsc_ctx* ctx, lrsc_batch* b, uint32_t* seed_count, lrsc_seed* seeds, uint64_t cap,
                                uint64_t* n_seeds, int8_t* attribute)
{
    if(!ctx || !b || b->ctx != ctx || !seed_count || !n_seeds) return fail(LRSC_ERR_ARG, "null / foreign batch");
    if(!b->seeds_done) return fail(LRSC_ERR_ARG, "call lrsc_batch_find_seeds first");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipMemcpy(seed_count, b->d_seed_count, (size_t)b->n_reads * sizeof(uint32_t), hipMemcpyDeviceToHost));
    std::vector<uint64_t> off(b->n_reads + 1);
    HIP_TRY(hipMemcpy(off.data(), b->d_off, off.size() * sizeof(uint64_t), hipMemcpyDeviceToHost));
    uint64_t total = 0;
    for(uint32_t r = 0; r < b->n_reads; ++r) total += seed_count[r];
    *n_seeds = total;
    if(total > cap) return fail(LRSC_ERR_CAPACITY, "seed buffer too small");
    if(seeds) {
        static_assert(sizeof(lrsc_seed) == kSeedInts * sizeof(int32_t), "lrsc_seed layout");
        uint64_t w = 0;
        for(uint32_t r = 0; r < b->n_reads; ++r) {
            if(seed_count[r] == 0) continue;
            const uint64_t slab = seed_slab(off[r], r, b->min_k);
            HIP_TRY(hipMemcpy(seeds + w, b->d_seeds + slab * kSeedInts, (size_t)seed_count[r] * sizeof(lrsc_seed),
                              hipMemcpyDeviceToHost));
            w += seed_count[r];
        }
    }
    if(attribute) HIP_TRY(hipMemcpy(attribute, b->d_attr, b->total_bases, hipMemcpyDeviceToHost));
    return LRSC_OK;
}

extern "C" int lrsc_find_seeds(lrsc_ctx* ctx, const char* reads, const uint64_t* read_off, uint32_t n_reads,
                               uint32_t* seed_count, lrsc_seed* seeds, uint64_t cap, uint64_t* n_seeds, int8_t* attribute)
{
    lrsc_batch* b = nullptr;
    int st = lrsc_batch_create(ctx, reads, read_off, n_reads, &b);
    if(st != LRSC_OK) return st;
    st = lrsc_batch_find_seeds(ctx, b);
    if(st == LRSC_OK) st = lrsc_batch_seeds(ctx, b, seed_count, seeds, cap, n_seeds, attribute);
    lrsc_batch_destroy(b);
    return st;
}

// ---- --debugseed / --onlyseed diagnostics ------------------------------------------------------------------
extern "C" int lrsc_batch_set_debug(lrsc_batch* b, int flags)
{
    if(!b) return fail(LRSC_ERR_ARG, "null batch");
    if(flags & ~(LRSC_DEBUG_OUTCASTS | LRSC_DEBUG_WALKS | LRSC_DEBUG_RATIO)) return fail(LRSC_ERR_ARG, "unknown debug flag");
    if(b->d_seeds) return fail(LRSC_ERR_ARG, "lrsc_batch_set_debug must precede lrsc_batch_find_seeds");
    b->debug_flags = flags;
    return LRSC_OK;
}

extern "C" int lrsc_batch_outcast_seeds(lrsc_ctx* ctx, lrsc_batch* b, uint32_t* outcast_count, lrsc_seed* seeds, uint64_t cap,
                                        uint64_t* n_seeds)
{
    if(!ctx || !b || b->ctx != ctx || !outcast_count || !n_seeds) return fail(LRSC_ERR_ARG, "null / foreign batch");
    if(!b->seeds_done || !b->d_outcasts) return fail(LRSC_ERR_ARG, "needs LRSC_DEBUG_OUTCASTS and lrsc_batch_find_seeds");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipMemcpy(outcast_count, b->d_outcast_count, (size_t)b->n_reads * sizeof(uint32_t), hipMemcpyDeviceToHost));
    std::vector<uint64_t> off(b->n_reads + 1);
    HIP_TRY(hipMemcpy(off.data(), b->d_off, off.size() * sizeof(uint64_t), hipMemcpyDeviceToHost));
    uint64_t total = 0;
    for(uint32_t r = 0; r < b->n_reads; ++r) total += outcast_count[r];
    *n_seeds = total;
    if(total > cap) return fail(LRSC_ERR_CAPACITY, "seed buffer too small");
    uint64_t w = 0;
    for(uint32_t r = 0; seeds && r < b->n_reads; ++r) {
        if(outcast_count[r] == 0) continue;
        const uint64_t slab = seed_slab(off[r], r, b->min_k);
        HIP_TRY(hipMemcpy(seeds + w, b->d_outcasts + slab * kSeedInts, (size_t)outcast_count[r] * sizeof(lrsc_seed), hipMemcpyDeviceToHost));
        w += outcast_count[r];
    }
    return LRSC_OK;
}

extern "C" int lrsc_batch_repeat_ratio(lrsc_ctx* ctx, lrsc_batch* b, float* ratio)
{
    if(!ctx || !b || b->ctx != ctx || !ratio) return fail(LRSC_ERR_ARG, "null / foreign batch");
    if(!b->seeds_done || !b->d_ratio) return fail(LRSC_ERR_ARG, "needs LRSC_DEBUG_RATIO and lrsc_batch_find_seeds");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipMemcpy(ratio, b->d_ratio, b->total_bases * sizeof(float), hipMemcpyDeviceToHost));
    return LRSC_OK;
}

extern "C" int lrsc_batch_walk_log(lrsc_ctx* ctx, lrsc_batch* b, uint8_t* log, uint64_t cap)
{
    if(!ctx || !b || b->ctx != ctx || !log) return fail(LRSC_ERR_ARG, "null / foreign batch");
    if(!b->walk_log_done || !b->d_walk_log) return fail(LRSC_ERR_ARG, "needs LRSC_DEBUG_WALKS and lrsc_batch_correct");
    HIP_TRY(hipSetDevice(ctx->device));
    std::vector<uint32_t> count(b->n_reads);
    std::vector<uint64_t> off(b->n_reads + 1);
    HIP_TRY(hipMemcpy(count.data(), b->d_seed_count, count.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(off.data(), b->d_off, off.size() * sizeof(uint64_t), hipMemcpyDeviceToHost));
    std::vector<uint8_t> all(b->seed_cap);
    HIP_TRY(hipMemcpy(all.data(), b->d_walk_log, b->seed_cap, hipMemcpyDeviceToHost));
    uint64_t w = 0;
    for(uint32_t r = 0; r < b->n_reads; ++r) {
        if(w + count[r] > cap) return fail(LRSC_ERR_CAPACITY, "walk log buffer too small");
        if(count[r]) std::memcpy(log + w, all.data() + seed_slab(off[r], r, b->min_k), count[r]);
        w += count[r];
    }
    return LRSC_OK;
}

// ---------------------------------------------------------------------------------------
// FM-extend
// ---------------------------------------------------------------------------------------
static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

extern "C" int lrsc_extend_walks(lrsc_ctx* ctx, const char* seq, uint64_t seq_len, const lrsc_walk_desc* walks, uint32_t n,
                                 lrsc_walk_result* results, char* out_arena, uint64_t arena_cap, uint64_t* arena_used)
{
    if(!ctx || (!walks && n) || (!results && n) || !arena_used) return fail(LRSC_ERR_ARG, "null");
    *arena_used = 0;
    if(n == 0) return LRSC_OK;
    const lrsc_params& p = ctx->params;
    if(p.max_leaves < 1 || p.max_leaves > 32) return fail(LRSC_ERR_UNSUPPORTED, "max_leaves must be 1..32");
    if(p.idmer_len < 5 || p.idmer_len > 16) return fail(LRSC_ERR_UNSUPPORTED, "idmer_len must be 5..16");
    if(p.min_kmer_len < p.idmer_len || p.min_kmer_len > 62) return fail(LRSC_ERR_UNSUPPORTED, "min_kmer_len out of range");
    HIP_TRY(hipSetDevice(ctx->device));
    const bool wide = ctx->fm.wide != 0;
    const size_t psz = wide ? 8 : 4;
    const size_t lbytes = leaf_bytes(wide);

    // ---- geometry, codes, workspace plan --------------------------------------------------------------
    std::vector<WalkWork> work(n);
    std::vector<uint64_t> q_off(n + 1, 0);
    uint64_t codes_total = 0, ws_total = 0, out_total = 0;
    for(uint32_t w = 0; w < n; ++w) {
        const lrsc_walk_desc& d = walks[w];
        if(d.init_kmer == 0 || d.src_len < d.init_kmer || d.init_kmer < (uint32_t)p.idmer_len || d.max_overlap + 1 > 62 ||
           d.init_kmer > 62)
            return fail(LRSC_ERR_ARG, "walk: init_kmer must satisfy idmer_len <= init_kmer <= src_len and max_overlap < 62");
        if(d.trg_len < (uint32_t)p.min_kmer_len) return fail(LRSC_ERR_ARG, "walk: target seed shorter than min_kmer_len");
        if(d.seq_off + (uint64_t)d.src_len + d.path_len + d.trg_len > seq_len) return fail(LRSC_ERR_ARG, "walk: sequence out of range");
        WalkWork& ww = work[w];
        ww.initk = d.init_kmer; ww.path_len = d.path_len; ww.trg_len = d.trg_len; ww.dis = d.dis;
        ww.max_overlap = d.max_overlap; ww.min_sa = d.min_sa_threshold;
        ww.lq = d.init_kmer + d.path_len + d.trg_len;
        if(ww.lq >= 65535) return fail(LRSC_ERR_UNSUPPORTED, "walk: query longer than 65534 bases");
        const double maxLength = (1.2 * (d.dis + 10)) + (double)(2 * (uint64_t)d.init_kmer);
        if(maxLength < 0 || maxLength > 1e6) return fail(LRSC_ERR_ARG, "walk: dis out of range");
        ww.pathw = (uint32_t)(((uint64_t)maxLength + 4 + 15) / 16 + 1);
        ww.codes_off = codes_total;
        codes_total += ww.lq;
        q_off[w + 1] = q_off[w] + ww.lq;
        const uint32_t n9 = ww.lq - (uint32_t)p.idmer_len + 1, n5 = ww.lq - 5 + 1;
        const uint32_t nT = d.trg_len - (uint32_t)p.min_kmer_len + 1;
        size_t o = 0;
        ww.o_item9f = (uint32_t)o; o += (size_t)n9 * sizeof(SortItem);
        ww.o_item9r = (uint32_t)o; o += (size_t)n9 * sizeof(SortItem);
        ww.o_term = (uint32_t)o;   o = align_up(o + (size_t)nT * 4 * psz, 16);
        ww.o_leaves = (uint32_t)o; o = align_up(o + (size_t)(32 + kMaxChildren) * lbytes, 16);
        ww.o_rings = (uint32_t)o;  o += (size_t)32 * 100 * sizeof(double);
        ww.o_results = (uint32_t)o; o += (size_t)kMaxResults * sizeof(WalkResultRec);
        ww.o_paths = (uint32_t)o;  o += (size_t)(32 + kMaxResults) * ww.pathw * 4;
        ww.o_next9f = (uint32_t)o; o += (size_t)n9 * 2;
        ww.o_next9r = (uint32_t)o; o += (size_t)n9 * 2;
        ww.o_head9 = (uint32_t)o;  o += 512 * 2;
        ww.o_head5 = (uint32_t)o;  o += 1024 * 2;
        ww.o_next5 = (uint32_t)o;  o += (size_t)n5 * 2;
        ww.o_flags5 = (uint32_t)o; o += n5;
        o = align_up(o, 64);
        if(o >= (1ull << 32)) return fail(LRSC_ERR_UNSUPPORTED, "walk workspace too large");
        ww.ws_off = ws_total;
        ws_total += o;
        ww.out_off = out_total;
        out_total += (size_t)ww.pathw * 4;
    }
    std::vector<uint8_t> codes(codes_total);
    for(uint32_t w = 0; w < n; ++w) {
        const lrsc_walk_desc& d = walks[w];
        const char* src = seq + d.seq_off + (d.src_len - d.init_kmer);     // beginningkmer = sourceSeed.substr(len - initk)
        uint8_t* dst = codes.data() + work[w].codes_off;
        for(uint32_t i = 0; i < work[w].lq; ++i) {
            const char c = src[i];
            uint8_t code;
            switch(c) { case 'A': code = 0; break; case 'C': code = 1; break; case 'G': code = 2; break; case 'T': code = 3; break;
                        default: return fail(LRSC_ERR_ARG, "sequence contains a base other than A,C,G,T"); }
            dst[i] = code;
        }
    }
    // launch order: long walks first, similar lengths share a wavefront
    std::vector<uint32_t> order(n);
    for(uint32_t i = 0; i < n; ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return walks[a].dis > walks[b].dis; });

    // pow() table of the constructor (.cpp:68-70), computed with the host libm like the reference does
    double freqs[101];
    for(int i = 0; i <= 100; ++i) freqs[i] = 0;
    for(int i = p.min_kmer_len; i <= 100; i++) freqs[i] = pow(1 - p.error_rate, i) * (size_t)p.pb_coverage;

    // ---- device buffers ------------------------------------------------------------------------------------
    DevBuf<uint8_t> d_codes, d_ws, d_outp;
    DevBuf<WalkWork> d_work;
    DevBuf<uint64_t> d_qoff;
    DevBuf<uint32_t> d_chunk, d_order;
    DevBuf<WalkOut> d_out;
    DevBuf<double> d_freqs;
    const uint64_t total_q = q_off[n];
    const uint64_t n_chunks = (total_q + (1ull << kChunkShift) - 1) >> kChunkShift;
    HIP_TRY(d_codes.reserve(codes_total));
    HIP_TRY(d_ws.reserve(ws_total));
    HIP_TRY(d_outp.reserve(out_total));
    HIP_TRY(d_work.reserve(n));
    HIP_TRY(d_qoff.reserve(n + 1));
    HIP_TRY(d_chunk.reserve(n_chunks));
    HIP_TRY(d_order.reserve(n));
    HIP_TRY(d_out.reserve(n));
    HIP_TRY(d_freqs.reserve(101));
    HIP_TRY(hipMemcpyAsync(d_codes.p, codes.data(), codes_total, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(d_work.p, work.data(), (size_t)n * sizeof(WalkWork), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(d_qoff.p, q_off.data(), (size_t)(n + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(d_order.p, order.data(), (size_t)n * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(d_freqs.p, freqs, sizeof(freqs), hipMemcpyHostToDevice, ctx->stream));
    hipError_t e = launch_chunk_table(d_qoff.p, n, total_q, d_chunk.p, ctx->stream);
    if(e != hipSuccess) return hip_fail(e, "chunk_table");

    ExtendArgs a{};
    a.codes = d_codes.p; a.work = d_work.p; a.q_off = d_qoff.p; a.chunk_walk = d_chunk.p; a.order = d_order.p;
    a.total_q = total_q; a.n_walks = n;
    a.workspace = d_ws.p; a.out_paths = d_outp.p; a.out = d_out.p;
    a.seed_size = (uint32_t)p.idmer_len; a.min_overlap = (uint32_t)p.min_kmer_len; a.max_leaves = (uint32_t)p.max_leaves;
    a.pb_coverage = (uint64_t)p.pb_coverage; a.pacbio_error_rate = p.error_rate;
    a.freqs_of_kmer_size = d_freqs.p;
    a.ctr = ctx->d_ctr;
    const int st = timed_launch(ctx, LRSC_K_EXTEND, [&]() {
        hipError_t e2 = launch_walk_prepare(ctx->fm, a, ctx->stream);
        if(e2 == hipSuccess) e2 = launch_walk_extend(ctx->fm, a, ctx->stream);
        return e2;
    });
    if(st != LRSC_OK) return st;

    std::vector<WalkOut> out(n);
    std::vector<uint32_t> outp(out_total / 4);
    HIP_TRY(hipMemcpy(out.data(), d_out.p, (size_t)n * sizeof(WalkOut), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(outp.data(), d_outp.p, out_total, hipMemcpyDeviceToHost));
    uint64_t used = 0;
    for(uint32_t w = 0; w < n; ++w) {
        if(out[w].code <= LRSC_WALK_ERR_CHILDREN) return fail(LRSC_ERR_LIMIT, "walk: internal frontier/result capacity exceeded");
        results[w].code = out[w].code; results[w].steps = out[w].steps; results[w].out_off = used; results[w].out_len = 0; results[w].pad = 0;
        if(out[w].code > 0) {
            const lrsc_walk_desc& d = walks[w];
            const uint32_t tail_from = out[w].match_i + (uint32_t)p.min_kmer_len;
            const uint32_t tail = d.trg_len > (uint32_t)p.min_kmer_len && tail_from <= d.trg_len ? d.trg_len - tail_from : 0;
            const uint32_t len = out[w].path_len + tail;
            results[w].out_len = len;
            if(out_arena && used + len <= arena_cap) {
                char* dst = out_arena + used;
                const uint32_t* pw = outp.data() + work[w].out_off / 4;
                for(uint32_t i = 0; i < out[w].path_len; ++i) dst[i] = "ACGT"[(pw[i >> 4] >> (2 * (i & 15))) & 3u];
                const char* trg = seq + d.seq_off + d.src_len + d.path_len;
                for(uint32_t i = 0; i < tail; ++i) dst[out[w].path_len + i] = trg[tail_from + i];
            }
            used += len;
        }
    }
    *arena_used = used;
    if(used > arena_cap || (!out_arena && used)) return fail(LRSC_ERR_CAPACITY, "output arena too small");
    return LRSC_OK;
}

static int encode_acgt(const char* seq, uint64_t n, std::vector<uint8_t>& codes)
{
    codes.resize(n);
    for(uint64_t i = 0; i < n; ++i) {
        switch(seq[i]) {
            case 'A': codes[i] = 0; break; case 'C': codes[i] = 1; break; case 'G': codes[i] = 2; break; case 'T': codes[i] = 3; break;
            default: return fail(LRSC_ERR_ARG, "sequence contains a base other than A,C,G,T");
        }
    }
    return LRSC_OK;
}

static uint32_t dp_wave_count(const lrsc_ctx* ctx)
{
    int cus = 256;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device);
    return (uint32_t)cus * 8u;
}

// The DP stage for a set of requests whose queries are already on the device: seeds -> (chunked by memory)
// retrieve -> align -> MSA.  Results stay on the device (d_msa[i], consensus codes at d_cons + reqs[i].cons_off).
struct DpStage {
    DevBuf<DpRequest> d_reqs;
    DevBuf<DpMsaOut> d_msa;
    DevBuf<uint8_t> d_cons, d_strings, d_ops, d_trace;
    DevBuf<DpJob> d_jobs;
    DevBuf<DpAlignOut> d_align;
    DevBuf<uint32_t> d_list;
    DevBuf<uint8_t> d_msa_ws, d_seq_ws;
    DevBuf<uint32_t> d_msa_ctr;
    uint64_t cons_total = 0, n_strings = 0;
    // the MSA size buckets of a round run concurrently on side streams (each bucket's launch ends with a tail of a few long
    // pile-ups; serialised, those tails cost more than the work)
    static constexpr int kSide = 8;
    hipStream_t side[kSide] = {};
    hipEvent_t side_done[kSide] = {};
    ~DpStage()
    {
        for(int i = 0; i < kSide; ++i) {
            if(side_done[i]) (void)hipEventDestroy(side_done[i]);
            if(side[i]) (void)hipStreamDestroy(side[i]);
        }
    }

    int run(lrsc_ctx* ctx, const uint8_t* d_query_codes, std::vector<DpRequest>& reqs)
    {
        const uint32_t n = (uint32_t)reqs.size();
        n_strings = 0; cons_total = 0;
        if(n == 0) return LRSC_OK;
        std::vector<uint8_t> too_long(n, 0);
        bool any_too_long = false;
        for(DpRequest& r : reqs) {
            if(r.k == 0 || r.lq < r.k) return fail(LRSC_ERR_ARG, "dp request: kmer_len must satisfy 1 <= kmer_len <= query length");
            if(r.coverage > 1000) return fail(LRSC_ERR_UNSUPPORTED, "dp request: coverage above 1000 (12-bit column counters)");
            r.max_len = (uint32_t)(size_t)(r.lq * 1.1 + 20);                 // LongReadOverlap.cpp:618
            r.str_cap = (std::max(r.max_len, r.k) + 3) & ~3u;
            r.ops_cap = (r.lq + r.str_cap + 1 + 3) & ~3u;
            r.cons_cap = dp_cons_capacity(r.lq);
            r.w_cols = dp_msa_columns(r.lq);
            r.cons_off = cons_total;
            cons_total += r.cons_cap;
            // beyond the alignment kernel's LDS staging (query + one retrieved string: about 30 kb of query): its alignments go through the
            // global-workspace variant of the kernel in a second launch
            if(dp_align_stage_bytes(r.lq, r.str_cap) > kDpAlignLdsCap) too_long[(size_t)(&r - reqs.data())] = 1;
        }
        HIP_TRY(d_reqs.reserve(n));
        HIP_TRY(d_msa.reserve(n));
        HIP_TRY(d_cons.reserve(cons_total));
        HIP_TRY(hipMemcpyAsync(d_reqs.p, reqs.data(), (size_t)n * sizeof(DpRequest), hipMemcpyHostToDevice, ctx->stream));
        DpPipeArgs a{};
        a.codes = d_query_codes; a.reqs = d_reqs.p; a.n_reqs = n; a.ctr = ctx->d_ctr;
        { const char* e = std::getenv("LRSC_MSA_BATCH"); a.row_batch = !(e && e[0] == '0'); }
        int st = timed_launch(ctx, LRSC_K_LF, [&]() { return launch_dp_seeds(ctx->fm, a, ctx->stream); });
        if(st != LRSC_OK) return st;
        HIP_TRY(hipMemcpy(reqs.data(), d_reqs.p, (size_t)n * sizeof(DpRequest), hipMemcpyDeviceToHost));

        uint64_t budget = 16ull << 30;
        if(const char* e = std::getenv("LRSC_DP_CHUNK_MB")) budget = std::max<uint64_t>(1, (uint64_t)std::atoll(e)) << 20;
        const uint32_t n_waves = dp_wave_count(ctx);
        uint32_t begin = 0;
        while(begin < n) {
            uint32_t end = begin, max1 = 1, max2 = 1, lds = 0, max1_long = 0, max2_long = 0;
            uint64_t jobs_long = 0;
            uint64_t jobs = 0, sbytes = 0, obytes = 0;
            while(end < n) {
                DpRequest& r = reqs[end];
                r.n_str = r.cnt[0] + r.cnt[1] + r.cnt[2] + r.cnt[3];
                any_too_long = any_too_long || too_long[end];
                const uint64_t sb = (uint64_t)r.n_str * r.str_cap, ob = (uint64_t)r.n_str * r.ops_cap;
                if(end > begin && sbytes + obytes + sb + ob + (jobs + r.n_str) * (sizeof(DpJob) + sizeof(DpAlignOut)) > budget) break;
                r.job_first = jobs; r.str_off = sbytes; r.ops_off = obytes;
                jobs += r.n_str; sbytes += sb; obytes += ob;
                if(!too_long[end]) { max1 = std::max(max1, r.lq); max2 = std::max(max2, r.str_cap); }
                else { max1_long = std::max(max1_long, r.lq); max2_long = std::max(max2_long, r.str_cap); jobs_long += r.n_str; }
                lds = std::max(lds, dp_msa_lds_bytes(r.w_cols, r.lq, r.str_cap, r.ops_cap, r.n_str));
                ++end;
            }
            if(jobs >= (1ull << 32)) return fail(LRSC_ERR_UNSUPPORTED, "dp chunk: too many alignments");
            const uint32_t nc = end - begin;
            HIP_TRY(hipMemcpyAsync(d_reqs.p + begin, reqs.data() + begin, (size_t)nc * sizeof(DpRequest), hipMemcpyHostToDevice, ctx->stream));
            HIP_TRY(d_strings.reserve(std::max<uint64_t>(sbytes, 64)));
            HIP_TRY(d_ops.reserve(std::max<uint64_t>(obytes, 64)));
            HIP_TRY(d_jobs.reserve(std::max<uint64_t>(jobs, 1)));
            HIP_TRY(d_align.reserve(std::max<uint64_t>(jobs, 1)));
            DpPipeArgs c = a;
            c.reqs = d_reqs.p + begin; c.n_reqs = nc; c.n_jobs = jobs;
            c.strings = d_strings.p; c.jobs = d_jobs.p; c.align = d_align.p; c.ops = d_ops.p;
            c.cons = d_cons.p; c.msa = d_msa.p + begin; c.lds_bytes = lds;
            st = timed_launch(ctx, LRSC_K_LF, [&]() { return launch_dp_retrieve(ctx->fm, c, ctx->stream); });
            if(st != LRSC_OK) return st;
            if(jobs) {
                DpAlignArgs al{};
                al.codes = d_query_codes; al.strings = d_strings.p; al.jobs = d_jobs.p; al.n_jobs = (uint32_t)jobs;
                al.band_width = 200; al.match_score = 1; al.gap_penalty = -1; al.mismatch_penalty = -8;   // LongReadOverlap.cpp:635-643
                al.ops = d_ops.p; al.out = d_align.p; al.max_s1 = max1; al.max_s2 = max2; al.reqs = c.reqs;
                al.trace_stride = (uint64_t)(max1 + 17) * kDpTraceStride;
                // 8 wavefronts per SIMD hide the scan's cross-lane latency; the traceback scratch is capped at 4 GB
                uint64_t nw64 = std::min<uint64_t>((uint64_t)n_waves * 4, jobs);
                nw64 = std::max<uint64_t>(1, std::min<uint64_t>(nw64, (4ull << 30) / al.trace_stride));
                const uint32_t nw = (uint32_t)nw64;
                HIP_TRY(d_trace.reserve(al.trace_stride * nw));
                al.trace = d_trace.p;
                al.lds_cap = kDpAlignLdsCap;
                st = timed_launch(ctx, LRSC_K_DP, [&]() { return launch_dp_align(al, nw, ctx->stream); });
                if(st != LRSC_OK) return st;
                if(jobs_long) {
                    // the alignments of requests beyond the LDS stage: same kernel, sequences staged in a global slice per wavefront
                    DpAlignArgs gl = al;
                    gl.max_s1 = max1_long; gl.max_s2 = max2_long; gl.only_long = 1;
                    gl.trace_stride = (uint64_t)(max1_long + 17) * kDpTraceStride;
                    gl.seq_ws_stride = (dp_align_stage_bytes(max1_long, max2_long) + 255) & ~255ull;
                    const uint32_t nwl = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(std::min<uint64_t>(jobs_long, 1024), (4ull << 30) / gl.trace_stride));
                    HIP_TRY(d_trace.reserve(gl.trace_stride * nwl));
                    HIP_TRY(d_seq_ws.reserve(gl.seq_ws_stride * nwl));
                    gl.trace = d_trace.p; gl.seq_ws = d_seq_ws.p;
                    st = timed_launch(ctx, LRSC_K_DP, [&]() { return launch_dp_align(gl, nwl, ctx->stream); });
                    if(st != LRSC_OK) return st;
                }
            }
            if(std::getenv("LRSC_CORRECT_PROFILE") && begin == 0 && jobs) {
                std::vector<DpAlignOut> ao(jobs);
                (void)hipMemcpy(ao.data(), d_align.p, jobs * sizeof(DpAlignOut), hipMemcpyDeviceToHost);
                double tf = 0, tt = 0, cols = 0, na = 0;
                for(const DpAlignOut& o : ao) if(!o.skipped) { tf += o.t_fill; tt += o.t_trace; cols += o.total_columns; na += 1; }
                std::fprintf(stderr, "[lrsc] align: %llu jobs (%.0f aligned), %.0f columns avg, per job %.0f ticks fill + %.0f ticks traceback; lds %u B, %u waves\n",
                             (unsigned long long)jobs, na, cols / std::max(na, 1.0), tf / std::max(na, 1.0), tt / std::max(na, 1.0),
                             ((max1 + 2 + 3) & ~3u) + max2 + 8, (unsigned)std::min<uint64_t>((uint64_t)n_waves * 4, jobs));
            }
            if(std::getenv("LRSC_DP_DEBUG")) {
                std::vector<DpAlignOut> ao(jobs);
                std::vector<DpJob> jj(jobs);
                (void)hipMemcpy(ao.data(), d_align.p, jobs * sizeof(DpAlignOut), hipMemcpyDeviceToHost);
                (void)hipMemcpy(jj.data(), d_jobs.p, jobs * sizeof(DpJob), hipMemcpyDeviceToHost);
                for(uint32_t i = begin; i < end && i < begin + 3; ++i) {
                    const DpRequest& r = reqs[i];
                    std::fprintf(stderr, "[dp] req %u lq %u k %u cov %u cnt %u %u %u %u rows %llu %llu %llu %llu n_str %u max_len %u\n", i, r.lq, r.k,
                                 r.coverage, r.cnt[0], r.cnt[1], r.cnt[2], r.cnt[3], (unsigned long long)r.row_lo[0], (unsigned long long)r.row_lo[1],
                                 (unsigned long long)r.row_lo[2], (unsigned long long)r.row_lo[3], r.n_str, r.max_len);
                    for(uint32_t s = 0; s < r.n_str; ++s) {
                        const DpAlignOut& o = ao[r.job_first + s];
                        std::fprintf(stderr, "[dp]   str %u len %u mode %u skipped %u accept %u cols %d edit %d m0 %d-%d m1 %d-%d nops %u\n", s,
                                     jj[r.job_first + s].s2_len, jj[r.job_first + s].mode, o.skipped, o.accept, o.total_columns, o.edit_distance,
                                     o.m0s, o.m0e, o.m1s, o.m1e, o.n_ops);
                    }
                }
            }
            // multiple alignments: one launch per LDS-size bucket (a wide pile-up must not cut everyone's occupancy), a
            // global-memory variant for the few that exceed 160 KB; a pile-up that opened more gap columns than its
            // capacity is redone with twice the columns.
            std::vector<DpMsaOut> mo(nc);
            std::vector<uint32_t> todo(nc), list;
            for(uint32_t i = 0; i < nc; ++i) todo[i] = i;
            HIP_TRY(d_list.reserve(nc));
            while(!todo.empty()) {
                static const uint32_t kBuckets[] = {8u << 10, 12u << 10, 16u << 10, 20u << 10, 24u << 10, 32u << 10, 40u << 10, 80u << 10, 160u << 10, 0xFFFFFFFFu};
                const bool force_global = std::getenv("LRSC_MSA_FORCE_GLOBAL") != nullptr;     // test hook for the global-workspace variant
                // one launch per bucket, all in flight together
                struct Launch { DpPipeArgs args; };
                std::vector<Launch> launches;
                std::vector<uint32_t> all_lists;
                uint32_t lo = 0;
                uint64_t ws_bytes = 0;
                for(uint32_t bk : kBuckets) {
                    if(force_global && bk != 0xFFFFFFFFu) continue;
                    const size_t first = all_lists.size();
                    uint32_t need_max = 0;
                    for(uint32_t i : todo) {
                        const DpRequest& r = reqs[begin + i];
                        const uint32_t need = dp_msa_lds_bytes(r.w_cols, r.lq, r.str_cap, r.ops_cap, r.n_str);
                        if(need > lo && need <= bk) { all_lists.push_back(i); need_max = std::max(need_max, need); }
                    }
                    lo = bk;
                    if(all_lists.size() == first) continue;
                    // a launch hands its requests to the wavefronts round-robin: the biggest pile-ups (rows x columns) first, so that the
                    // launch does not end on one of them
                    std::sort(all_lists.begin() + (std::ptrdiff_t)first, all_lists.end(), [&](uint32_t x, uint32_t y) {
                        const DpRequest& rx = reqs[begin + x]; const DpRequest& ry = reqs[begin + y];
                        const uint64_t wx = (uint64_t)rx.lq * rx.n_str, wy = (uint64_t)ry.lq * ry.n_str;
                        return wx != wy ? wx > wy : x < y;
                    });
                    Launch L;
                    L.args = c;
                    L.args.req_list = reinterpret_cast<const uint32_t*>(first);          // offset for now, pointer after the upload
                    L.args.n_list = (uint32_t)(all_lists.size() - first);
                    L.args.lds_bytes = (need_max + 15) & ~15u;
                    if(bk == 0xFFFFFFFFu) {
                        ws_bytes = (uint64_t)L.args.lds_bytes * dp_msa_waves(L.args, true);
                        L.args.msa_ws = reinterpret_cast<uint8_t*>(1);                     // marker, set below
                    }
                    launches.push_back(L);
                }
                HIP_TRY(d_list.reserve(std::max<size_t>(all_lists.size(), 1)));
                if(ws_bytes) HIP_TRY(d_msa_ws.reserve(ws_bytes));
                HIP_TRY(hipMemcpyAsync(d_list.p, all_lists.data(), all_lists.size() * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
                HIP_TRY(d_msa_ctr.reserve(std::max<size_t>(launches.size(), 16)));
                HIP_TRY(hipMemsetAsync(d_msa_ctr.p, 0, launches.size() * sizeof(uint32_t), ctx->stream));
                for(size_t j = 0; j < launches.size(); ++j) {
                    Launch& L = launches[j];
                    L.args.req_list = d_list.p + reinterpret_cast<size_t>(L.args.req_list);
                    if(L.args.msa_ws) L.args.msa_ws = d_msa_ws.p;
                    L.args.work_ctr = d_msa_ctr.p + j;
                }
                for(int i = 0; i < kSide; ++i) {
                    if(!side[i]) HIP_TRY(hipStreamCreateWithFlags(&side[i], hipStreamNonBlocking));
                    if(!side_done[i]) HIP_TRY(hipEventCreateWithFlags(&side_done[i], hipEventDisableTiming));
                }
                st = timed_launch(ctx, LRSC_K_MSA, [&]() -> hipError_t {
                    // ctx->ev0 was just recorded on ctx->stream: the side streams start after it (and after the list upload)
                    hipError_t e = hipSuccess;
                    for(size_t j = 0; j < launches.size() && e == hipSuccess; ++j) {
                        hipStream_t sj = side[j % kSide];
                        e = hipStreamWaitEvent(sj, ctx->ev0, 0);
                        if(e == hipSuccess) e = launch_dp_msa(launches[j].args, sj);
                    }
                    for(int i = 0; i < kSide && e == hipSuccess; ++i) {
                        e = hipEventRecord(side_done[i], side[i]);
                        if(e == hipSuccess) e = hipStreamWaitEvent(ctx->stream, side_done[i], 0);
                    }
                    return e;
                });
                if(st != LRSC_OK) return st;
                HIP_TRY(hipMemcpy(mo.data(), d_msa.p + begin, (size_t)nc * sizeof(DpMsaOut), hipMemcpyDeviceToHost));
                list.clear();
                for(uint32_t i : todo) {
                    if(mo[i].error != 1) continue;                       // 0 = done; 2 = consensus beyond its capacity: stays an error of this request
                    DpRequest& r = reqs[begin + i];
                    if(r.w_cols > 64u * (r.lq + 128)) continue;           // gives up on this pile-up: the request keeps its error
                    r.w_cols *= 2;
                    list.push_back(i);
                }
                if(std::getenv("LRSC_CORRECT_PROFILE") && begin == 0) {
                    double kt = 0, ks = 0, ki = 0, ni = 0, rows = 0, walked = 0;
                    for(uint32_t i = 0; i < nc; ++i) { kt += mo[i].kc_total; ks += mo[i].kc_stage; ki += mo[i].kc_insert; ni += mo[i].n_insert; rows += mo[i].n_rows; walked += mo[i].pad; }
                    std::fprintf(stderr, "[lrsc] msa: %u requests, %.1f rows avg (%.2f by the step walk), %.0f insertions avg, per request %.0f k-ticks (staging %.0f, insertions %.0f), redo %zu\n",
                                 nc, rows / nc, walked / nc, ni / nc, kt / nc, ks / nc, ki / nc, list.size());
                }
                todo = list;
                if(!todo.empty())
                    HIP_TRY(hipMemcpyAsync(d_reqs.p + begin, reqs.data() + begin, (size_t)nc * sizeof(DpRequest), hipMemcpyHostToDevice, ctx->stream));
            }
            n_strings += jobs;
            begin = end;
        }
        (void)any_too_long;
        return LRSC_OK;
    }
};

// Device buffers of lrsc_batch_correct, kept in the ctx between calls (grow-only): hipMalloc / hipFree synchronise the whole
// device, which serialises contexts that correct sub-batches concurrently on one GPU and costs every call of a loop.
struct CorrectScratch {
    DevBuf<ReadPlan> d_plan;
    DevBuf<uint32_t> d_pieces;
    DevBuf<uint8_t> d_codes_out;
    DevBuf<uint64_t> d_dst_off;
    DevBuf<char> d_dst;
    DevBuf<double> d_freqs;
    DpStage stage;
    struct WpScratch* wp = nullptr;          // buffers of the walk-parallel flow
};
struct WpScratch;
static void free_wp_scratch(WpScratch* w);
static void free_correct_scratch(CorrectScratch* cs) { if(cs) free_wp_scratch(cs->wp); delete cs; }

// ---------------------------------------------------------------------------------------
// the walk-parallel flow (wp.h / wp.hip): default implementation of lrsc_batch_correct
// ---------------------------------------------------------------------------------------
// Bump allocator over grow-only device chunks: results that later rounds still read (queries, result paths, DP consensus)
// live here until the read range is done.  reset() keeps the chunks, so a loop over same-shaped batches stops allocating.
struct DevArena {
    std::vector<DevBuf<uint8_t>*> chunks;
    size_t cur = 0, off = 0;
    ~DevArena() { for(auto* c : chunks) delete c; }
    void reset() { cur = 0; off = 0; }
    hipError_t alloc(size_t bytes, uint8_t** out)
    {
        bytes = (std::max<size_t>(bytes, 1) + 255) & ~(size_t)255;
        while(cur < chunks.size()) {
            if(chunks[cur]->cap - off >= bytes) { *out = chunks[cur]->p + off; off += bytes; return hipSuccess; }
            // an empty chunk that is too small is replaced rather than skipped (grow-only, few chunks)
            if(off == 0) { hipError_t e = chunks[cur]->reserve(bytes + bytes / 8); if(e != hipSuccess) return e; continue; }
            ++cur; off = 0;
        }
        auto* c = new(std::nothrow) DevBuf<uint8_t>();
        if(!c) return hipErrorOutOfMemory;
        hipError_t e = c->reserve(std::max<size_t>(bytes + bytes / 8, 64u << 20));
        if(e != hipSuccess) { delete c; return e; }
        chunks.push_back(c);
        cur = chunks.size() - 1;
        *out = c->p; off = bytes;
        return hipSuccess;
    }
};

struct WpScratch {
    DevBuf<WpReadWork> d_work;
    DevBuf<WpRead> d_reads;
    DevBuf<WpSlot> d_slots;
    DevBuf<uint64_t> d_sz;                 // three arrays of (entries + 1)
    DevBuf<uint32_t> d_key, d_key_tmp, d_list, d_list_tmp, d_small;   // d_small: plan_stats[4], queue, n_dp_items, n_req
    DevBuf<WpDpItem> d_items, d_items2, d_items3;
    DevBuf<DevCounters> d_ctr2;
    hipEvent_t ev_side_t0 = nullptr, ev_side_t1 = nullptr;
    DevBuf<WpRequest> d_req;
    DevBuf<uint8_t> d_prep, d_lane, d_lane_side, d_lane_side2, d_ctx[2];
    hipStream_t side[2] = {nullptr, nullptr};
    hipEvent_t ev_side[2] = {nullptr, nullptr}, ev_ready = nullptr;
    DevBuf<unsigned long long> d_prof;
    DevBuf<uint8_t> d_coop, d_coop_side;
    DevBuf<WpSched> d_sched;
    DevBuf<uint32_t> d_sched_lists[2];
    DevArena persist;
    void* cub_tmp = nullptr;
    size_t cub_cap = 0;
    ~WpScratch()
    {
        if(cub_tmp) (void)hipFree(cub_tmp);
        for(int i = 0; i < 2; ++i) {
            if(side[i]) { (void)hipStreamSynchronize(side[i]); (void)hipStreamDestroy(side[i]); }
            if(ev_side[i]) (void)hipEventDestroy(ev_side[i]);
        }
        if(ev_ready) (void)hipEventDestroy(ev_ready);
        if(ev_side_t0) (void)hipEventDestroy(ev_side_t0);
        if(ev_side_t1) (void)hipEventDestroy(ev_side_t1);
    }
};

static void free_wp_scratch(WpScratch* w) { delete w; }

static int batch_correct_wp(lrsc_ctx* ctx, lrsc_batch* b, lrsc_read_result* res, uint64_t* piece_off, uint64_t piece_cap, char* out,
                            uint64_t out_cap, uint64_t* n_pieces_out, uint64_t* out_used)
{
    const lrsc_params& p = ctx->params;
    const uint32_t n = b->n_reads;
    const bool wide = ctx->fm.wide != 0;
    const uint32_t psz = wide ? 8 : 4;
    const uint32_t lbytes = (uint32_t)leaf_bytes(wide);
    const bool verbose = std::getenv("LRSC_CORRECT_PROFILE") != nullptr;

    double freqs[101];
    for(int i = 0; i <= 100; ++i) freqs[i] = 0;
    for(int i = p.min_kmer_len; i <= 100; i++) freqs[i] = pow(1 - p.error_rate, i) * (size_t)p.pb_coverage;

    if(!ctx->cs) ctx->cs = new(std::nothrow) CorrectScratch();
    if(!ctx->cs) return fail(LRSC_ERR_NOMEM, "correct scratch");
    CorrectScratch& cs = *ctx->cs;
    if(!cs.wp) cs.wp = new(std::nothrow) WpScratch();
    if(!cs.wp) return fail(LRSC_ERR_NOMEM, "correct scratch");
    WpScratch& ws = *cs.wp;
    HIP_TRY(cs.d_plan.reserve(n));
    HIP_TRY(cs.d_freqs.reserve(101));
    HIP_TRY(hipMemcpyAsync(cs.d_freqs.p, freqs, sizeof(freqs), hipMemcpyHostToDevice, ctx->stream));

    // ---- per-read bounds (longest gap / query any walk of the read can have) -> output slots, skipped reads -----------
    WpArgs pa{};
    pa.codes = b->d_codes; pa.read_off = b->d_off; pa.seeds = b->d_seeds; pa.seed_count = b->d_seed_count;
    pa.n_reads = n; pa.min_k = b->min_k; pa.next_target = p.next_target;
    hipError_t e = launch_wp_bounds(pa, cs.d_plan.p, ctx->stream);
    if(e != hipSuccess) return hip_fail(e, "wp_bounds");
    std::vector<ReadPlan> plan(n);
    std::vector<uint32_t> seed_count(n);
    std::vector<uint64_t> off(n + 1);
    HIP_TRY(hipMemcpyAsync(plan.data(), cs.d_plan.p, (size_t)n * sizeof(ReadPlan), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(seed_count.data(), b->d_seed_count, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(off.data(), b->d_off, (size_t)(n + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));

    std::vector<WpReadWork> work(n);
    std::vector<int> skipped(n, 0);
    uint64_t out_total = 0, piece_total = 0, n_slots = 0;
    for(uint32_t r = 0; r < n; ++r) {
        WpReadWork& w = work[r];
        std::memset(&w, 0, sizeof(w));
        const uint64_t rlen = off[r + 1] - off[r];
        const uint32_t ns = seed_count[r];
        w.out_off = out_total; w.piece_off = piece_total; w.slot_first = n_slots;
        if(ns < 2) continue;                                          // nothing to correct: the read is discarded
        // every walk appends at most maxLength + 1 + |target| - initk characters, walks <= seeds, gaps sum to <= |read|
        // (a DP consensus can be longer than its query by the insertion columns it keeps: budget 2x the raw segment)
        const uint64_t cap = rlen + (uint64_t)((p.no_dp ? 1.2 : 2.0) * (double)rlen) + (uint64_t)ns * (2 * kMaxInitK + 16 + (p.no_dp ? 0 : 128)) + 64;
        if(cap >= (1ull << 32)) { skipped[r] = LRSC_READ_TOO_LONG; continue; }
        if(plan[r].lq_max >= 65535) { skipped[r] = LRSC_READ_WALK_QUERY_TOO_LONG; continue; }
        w.out_cap = (uint32_t)cap;
        w.piece_cap = p.split ? ns : 1;
        w.n_seeds = ns;
        out_total += ((uint64_t)w.out_cap + 15) & ~15ull;
        piece_total += w.piece_cap;
        n_slots += ns - 1;
    }
    if(n_slots >= (1ull << 32)) return fail(LRSC_ERR_UNSUPPORTED, "batch: more than 2^32 seed pairs (split the input)");

    HIP_TRY(ws.d_work.reserve(n));
    HIP_TRY(ws.d_reads.reserve(n));
    HIP_TRY(ws.d_slots.reserve(std::max<uint64_t>(n_slots, 1)));
    HIP_TRY(ws.d_small.reserve(16));
    HIP_TRY(ws.d_req.reserve(n));
    HIP_TRY(cs.d_codes_out.reserve(std::max<uint64_t>(out_total, 64)));
    HIP_TRY(cs.d_pieces.reserve(std::max<uint64_t>(piece_total, 1)));
    HIP_TRY(hipMemcpyAsync(ws.d_work.p, work.data(), (size_t)n * sizeof(WpReadWork), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemsetAsync(ws.d_reads.p, 0, (size_t)n * sizeof(WpRead), ctx->stream));

    WpArgs a{};
    a.codes = b->d_codes; a.read_off = b->d_off; a.seeds = b->d_seeds; a.seed_count = b->d_seed_count;
    a.n_reads = n; a.min_k = b->min_k;
    a.work = ws.d_work.p; a.reads = ws.d_reads.p; a.slots = ws.d_slots.p; a.n_slots = n_slots;
    a.seed_size = (uint32_t)p.idmer_len; a.min_overlap = (uint32_t)p.min_kmer_len; a.max_leaves = (uint32_t)p.max_leaves;
    a.start_kmer_len = p.start_kmer_len; a.next_target = p.next_target; a.split = p.split; a.no_dp = p.no_dp;
    a.pb_coverage = (uint64_t)p.pb_coverage; a.pacbio_error_rate = p.error_rate;
    a.freqs_of_kmer_size = cs.d_freqs.p;
    a.psz = psz; a.lbytes = lbytes;
    a.plan_stats = ws.d_small.p; a.queue = ws.d_small.p + 4; a.n_dp_items = ws.d_small.p + 5; a.n_req_out = ws.d_small.p + 6;
    a.req_out = ws.d_req.p; a.req_cap = n;
    a.out_codes = cs.d_codes_out.p; a.piece_start = cs.d_pieces.p;
    a.auto_dp = (!p.no_dp && p.next_target == 1) ? 1u : 0u;
    a.general_quorum_pct = 70; a.general_max_wait = 6;
    size_t dp_split_pct = 100;               // share of the bulk's failed walks in the first DP call when long-gap walks run beside it (100: theirs alone in the second)
    if(const char* ev = std::getenv("LRSC_WP_DP_SPLIT")) dp_split_pct = (size_t)std::min(100, std::max(10, std::atoi(ev)));
    if(const char* ev = std::getenv("LRSC_WP_GEN_QUORUM")) a.general_quorum_pct = (uint32_t)std::min(100, std::max(0, std::atoi(ev)));
    if(const char* ev = std::getenv("LRSC_WP_GEN_WAIT")) a.general_max_wait = (uint32_t)std::max(0, std::atoi(ev));
    a.ctr = ctx->d_ctr;
    if(b->debug_flags & LRSC_DEBUG_WALKS) {
        if(!b->d_walk_log) HIP_TRY(hipMalloc(reinterpret_cast<void**>(&b->d_walk_log), b->seed_cap));
        HIP_TRY(hipMemsetAsync(b->d_walk_log, 0, b->seed_cap, ctx->stream));
        a.walk_log = b->d_walk_log;
        b->walk_log_done = false;
    }

    if(verbose) {
        HIP_TRY(ws.d_prof.reserve(32));                                      // [16..32): the long-gap walks' side launch on its own
        HIP_TRY(hipMemsetAsync(ws.d_prof.p, 0, 32 * sizeof(unsigned long long), ctx->stream));
        a.prof = ws.d_prof.p;
    }
    int cus = 256;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device);
    uint32_t max_lanes = (uint32_t)cus * 4u * 2u * 64u;              // 2 wavefronts per SIMD (the extension kernels hold ~190 VGPRs)
    if(const char* ev = std::getenv("LRSC_WP_LANES")) max_lanes = (uint32_t)std::max(64, std::atoi(ev));
    uint64_t prep_budget = 24ull << 30, lane_budget = 32ull << 30;
    if(const char* ev = std::getenv("LRSC_WP_PREP_MB")) prep_budget = std::max<uint64_t>(1, (uint64_t)std::atoll(ev)) << 20;
    if(const char* ev = std::getenv("LRSC_WP_LANE_MB")) lane_budget = std::max<uint64_t>(1, (uint64_t)std::atoll(ev)) << 20;

    // side streams: the few walks across long gaps and the mid-size class run beside the bulk instead of before it (each is
    // latency-bound on its own: a walk is a chain of dependent steps)
    for(int i = 0; i < 2; ++i) {
        if(!ws.side[i]) HIP_TRY(hipStreamCreateWithFlags(&ws.side[i], hipStreamNonBlocking));
        if(!ws.ev_side[i]) HIP_TRY(hipEventCreateWithFlags(&ws.ev_side[i], hipEventDisableTiming));
    }
    if(!ws.ev_ready) HIP_TRY(hipEventCreateWithFlags(&ws.ev_ready, hipEventDisableTiming));

    uint32_t mid_stride = 64, long_mode = 0;
    if(const char* ev = std::getenv("LRSC_WP_LONG_MODE")) long_mode = (uint32_t)std::atoi(ev);
    bool coop_helpers = false;                       // experiment: run-ahead helper lanes in the one-walk-per-wavefront launches
    if(const char* ev = std::getenv("LRSC_WP_COOP")) coop_helpers = std::atoi(ev) != 0;
    uint32_t leaves_in_lds = 0;                      // measured: 77.5-77.7 vs 79.1-80.3 corrected Mbases/s with the leaves in LDS (the DP stage beside it wants the LDS)
    if(const char* ev = std::getenv("LRSC_WP_LEAVES_LDS")) leaves_in_lds = std::atoi(ev) != 0;
    uint32_t long_first_div = 8;
    if(const char* ev = std::getenv("LRSC_WP_SIDE_DIV")) long_first_div = (uint32_t)std::min(64, std::max(2, std::atoi(ev)));
    if(const char* ev = std::getenv("LRSC_WP_MID_STRIDE")) { const int v = std::atoi(ev); if(v == 1 || v == 2 || v == 4 || v == 8 || v == 16 || v == 32 || v == 64) mid_stride = (uint32_t)v; }
    // One launch of the one-kernel form (wp_extend_kernel: every lane runs both kinds of step) over `count` list entries;
    // stride 64 = one walk per wavefront
    bool reserve_side = false;
    uint64_t side_div = 2;                                                   // the side launch's share of the wavefront slots: 1 / side_div
    auto extend_range = [&](WpArgs x, const uint32_t* list, const WpRequest* reqs, uint32_t count, uint32_t pathw, hipStream_t st, int which, uint32_t stride) -> hipError_t {
        if(count == 0) return hipSuccess;
        const WpLaneLayout LL = wp_lane_layout(lbytes, pathw);
        // wavefront slots: the side launches (which != 0) and the bulk launch are persistent and share the device, so each gets a share
        // of the resident wavefronts -- a launch that fills every slot first would keep the others out until it ends
        const uint64_t slots = max_lanes / 64;                                        // resident wavefronts of these kernels
        const uint64_t side_part = std::max<uint64_t>(1, slots / side_div);
        const uint64_t share = which == 1 ? side_part : which == 2 ? slots / 4 : (reserve_side ? slots - side_part : slots);
        uint64_t lanes = std::min<uint64_t>(stride == 1 ? (((uint64_t)count + 63) & ~63ull) : count, share * 64 / stride);
        lanes = std::max<uint64_t>(1, std::min<uint64_t>(lanes, (lane_budget / (which ? 4 : 1)) / LL.total));
        if(stride == 1) lanes = std::max<uint64_t>(64, lanes & ~63ull);
        DevBuf<uint8_t>& buf = which == 2 ? ws.d_lane_side2 : which ? ws.d_lane_side : ws.d_lane;
        hipError_t e2 = buf.reserve(lanes * LL.total);
        if(e2 != hipSuccess) return e2;
        x.list = list; x.reqs = reqs; x.n_list = count;
        x.lane_ws = buf.p; x.lane_ws_bytes = LL.total; x.lane_pathw = pathw; x.n_lanes = (uint32_t)lanes; x.lane_stride = stride;
        x.leaves_in_lds = leaves_in_lds;
        x.queue = ws.d_small.p + 8 + which;
        e2 = hipMemsetAsync(x.queue, 0, sizeof(uint32_t), st);
        if(e2 != hipSuccess) return e2;
        if(coop_helpers && stride == 64) {
            // one walk per wavefront with run-ahead helper lanes (wp_extend_coop_kernel): a private leaf buffer per lane
            DevBuf<uint8_t>& cb = which ? ws.d_coop_side : ws.d_coop;
            e2 = cb.reserve(lanes * 64 * kWpCoopLeaves * lbytes);
            if(e2 != hipSuccess) return e2;
            return launch_wp_extend_coop(ctx->fm, x, cb.p, st);
        }
        return launch_wp_extend(ctx->fm, x, st);
    };

    // The two-class schedule (wp_fast_kernel / wp_general_kernel): contexts = walks in flight.  Up to two pools (result-path
    // classes) advance side by side, each on its own stream.
    bool use_sched = false;
    if(const char* ev = std::getenv("LRSC_WP_SCHED")) use_sched = std::atoi(ev) != 0;
    uint32_t sched_budget_fast = 32, sched_budget_general = 24, sched_quorum = 25;
    if(const char* ev = std::getenv("LRSC_WP_BUDGET_FAST")) sched_budget_fast = (uint32_t)std::max(1, std::atoi(ev));
    if(const char* ev = std::getenv("LRSC_WP_BUDGET_GENERAL")) sched_budget_general = (uint32_t)std::max(1, std::atoi(ev));
    if(const char* ev = std::getenv("LRSC_WP_QUORUM")) sched_quorum = (uint32_t)std::min(100, std::max(0, std::atoi(ev)));
    uint64_t sched_rounds = 0;
    struct Pool { WpSchedArgs sf, sg; uint32_t count = 0, lanes = 0; hipStream_t st = nullptr; bool done = true; };
    auto pool_setup = [&](Pool& P, int which, const uint32_t* list, uint32_t count, uint32_t pathw, uint64_t budget_bytes, hipStream_t st) -> hipError_t {
        P.count = count; P.st = st; P.done = count == 0;
        if(count == 0) return hipSuccess;
        const WpLaneLayout LL = wp_lane_layout(lbytes, pathw);
        WpSchedArgs sa{};
        sa.ctx_bytes = 64 + LL.total; sa.ctx_pathw = pathw;
        uint64_t n_ctx = std::min<uint64_t>(count, 2ull * max_lanes);
        n_ctx = std::max<uint64_t>(1, std::min<uint64_t>(n_ctx, budget_bytes / sa.ctx_bytes));
        sa.n_ctx = (uint32_t)n_ctx;
        hipError_t e2 = ws.d_ctx[which].reserve(n_ctx * sa.ctx_bytes);
        if(e2 == hipSuccess) e2 = ws.d_sched.reserve(2);
        if(e2 == hipSuccess) e2 = ws.d_sched_lists[which].reserve((size_t)kWpLists * n_ctx);
        if(e2 != hipSuccess) return e2;
        sa.sched = ws.d_sched.p + which; sa.fresh = list; sa.ctx_ws = ws.d_ctx[which].p;
        sa.quorum_pct = sched_quorum;
        e2 = launch_wp_sched_init(sa, ws.d_sched_lists[which].p, count, st);
        if(e2 != hipSuccess) return e2;
        P.lanes = (uint32_t)std::min<uint64_t>((n_ctx + 63) & ~63ull, max_lanes);
        P.sf = sa; P.sg = sa;
        P.sf.budget = sched_budget_fast; P.sg.budget = sched_budget_general;
        return hipSuccess;
    };
    auto pools_run = [&](const WpArgs& x, Pool* pools, int n_pools) -> hipError_t {
        for(uint32_t iter = 0;; ++iter) {
            bool any = false;
            for(int i = 0; i < n_pools; ++i) {
                Pool& P = pools[i];
                if(P.done) continue;
                any = true;
                // one round = fast kernel, general kernel (the two differ in their budget: steps / leaf-steps per pull)
                hipError_t e2 = launch_wp_sched_round(ctx->fm, x, P.sf, P.sg, P.lanes, P.lanes, P.st);
                if(e2 != hipSuccess) return e2;
                ++sched_rounds;
            }
            if(!any) break;
            if((iter & 3u) == 3u) {
                uint32_t fin[2] = {0, 0};
                for(int i = 0; i < n_pools; ++i)
                    if(!pools[i].done) {
                        hipError_t e2 = hipMemcpyAsync(&fin[i], &pools[i].sf.sched->finished, sizeof(uint32_t), hipMemcpyDeviceToHost, pools[i].st);
                        if(e2 != hipSuccess) return e2;
                    }
                for(int i = 0; i < n_pools; ++i)
                    if(!pools[i].done) {
                        hipError_t e2 = hipStreamSynchronize(pools[i].st);
                        if(e2 != hipSuccess) return e2;
                        if(fin[i] >= pools[i].count) pools[i].done = true;
                    }
                if(iter > 4000000u) return hipErrorLaunchFailure;
            }
        }
        return hipSuccess;
    };
    // everything on the side streams starts after what is on ctx->stream now, and ctx->stream goes on after them
    auto side_begin = [&]() -> hipError_t {
        hipError_t e2 = hipEventRecord(ws.ev_ready, ctx->stream);
        for(int i = 0; i < 2 && e2 == hipSuccess; ++i) e2 = hipStreamWaitEvent(ws.side[i], ws.ev_ready, 0);
        return e2;
    };
    auto side_join = [&]() -> hipError_t {
        hipError_t e2 = hipSuccess;
        for(int i = 0; i < 2 && e2 == hipSuccess; ++i) {
            e2 = hipEventRecord(ws.ev_side[i], ws.side[i]);
            if(e2 == hipSuccess) e2 = hipStreamWaitEvent(ctx->stream, ws.ev_side[i], 0);
        }
        return e2;
    };

    // ---- read ranges whose prepared tables fit the budget (about 40 bytes per query character + 6 KB per walk) -------------------
    DpStage& stage = cs.stage;
    std::vector<WpDpItem> items;
    std::vector<DpRequest> reqs;
    std::vector<WpRequest> hreq;
    std::vector<uint32_t> hlist;
    uint32_t r0 = 0;
    uint64_t rounds_total = 0;
    while(r0 < n) {
        uint32_t r1 = r0;
        uint64_t est = 0;
        while(r1 < n) {
            const uint64_t ns = work[r1].n_seeds;
            const uint64_t need = ns >= 2 ? 55 * (off[r1 + 1] - off[r1]) + ns * 4400 : 0;     // 39 B per query character + 16 B per target character + 3.3 KB per walk
            if(r1 > r0 && est + need > prep_budget) break;
            est += need;
            ++r1;
        }
        const uint64_t slot_base = work[r0].slot_first;
        const uint64_t slot_end = r1 < n ? work[r1].slot_first : n_slots;
        const uint32_t n_range = (uint32_t)(slot_end - slot_base);
        a.r0 = r0; a.r1 = r1; a.slot_base = slot_base;
        ws.persist.reset();
        if(n_range != 0)
        for(uint32_t round = 0;; ++round) {
            // entries of this round: every slot of the range (round 0) or what the stitch pass asked for
            uint32_t n_ent = n_range;
            a.list = nullptr; a.reqs = nullptr;
            if(round != 0) {
                uint32_t n_req = 0;
                HIP_TRY(hipStreamSynchronize(ctx->stream));                      // ctx->stream is non-blocking: plain hipMemcpy does not wait for it
                HIP_TRY(hipMemcpy(&n_req, a.n_req_out, sizeof(uint32_t), hipMemcpyDeviceToHost));
                if(n_req == 0) break;
                if(n_req > n) return fail(LRSC_ERR_LIMIT, "walk-parallel flow: request list overflow");
                hreq.resize(n_req);
                HIP_TRY(hipMemcpy(hreq.data(), ws.d_req.p, (size_t)n_req * sizeof(WpRequest), hipMemcpyDeviceToHost));
                std::sort(hreq.begin(), hreq.end(), [](const WpRequest& x, const WpRequest& y) { return x.slot < y.slot; });
                hlist.resize(n_req);
                for(uint32_t i = 0; i < n_req; ++i) hlist[i] = hreq[i].slot;
                n_ent = n_req;
            }
            HIP_TRY(ws.d_sz.reserve(3 * ((size_t)n_ent + 1)));
            HIP_TRY(ws.d_key.reserve(n_ent));
            HIP_TRY(ws.d_key_tmp.reserve(n_ent));
            HIP_TRY(ws.d_list.reserve(n_ent));
            HIP_TRY(ws.d_list_tmp.reserve(n_ent));
            HIP_TRY(ws.d_items.reserve(n_ent));
            a.sz_q = ws.d_sz.p; a.sz_prep = ws.d_sz.p + (n_ent + 1); a.sz_path = ws.d_sz.p + 2 * ((size_t)n_ent + 1);
            a.sort_key = ws.d_key.p;
            a.n_list = n_ent;
            a.dp_items = ws.d_items.p; a.dp_items_cap = n_ent;
            HIP_TRY(hipMemsetAsync(ws.d_sz.p, 0, 3 * ((size_t)n_ent + 1) * sizeof(uint64_t), ctx->stream));
            HIP_TRY(hipMemsetAsync(ws.d_small.p, 0, 16 * sizeof(uint32_t), ctx->stream));
            WpRequest* d_reqs_in = nullptr;
            if(round != 0) {
                // the request records move to the front half of a second buffer so that the stitch pass can write new ones
                HIP_TRY(ws.d_list_tmp.reserve(std::max<size_t>(n_ent, 2 * (size_t)n_ent)));
                d_reqs_in = reinterpret_cast<WpRequest*>(ws.d_list_tmp.p);
                HIP_TRY(hipMemcpyAsync(d_reqs_in, hreq.data(), (size_t)n_ent * sizeof(WpRequest), hipMemcpyHostToDevice, ctx->stream));
                HIP_TRY(hipMemcpyAsync(ws.d_list.p, hlist.data(), (size_t)n_ent * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
                a.list = ws.d_list.p; a.reqs = d_reqs_in;
            }
            e = launch_wp_plan(a, ctx->stream);
            if(e != hipSuccess) return hip_fail(e, "wp_plan");
            for(int j = 0; j < 3; ++j) {
                e = wp_scan(ws.d_sz.p + (size_t)j * (n_ent + 1), (uint64_t)n_ent + 1, &ws.cub_tmp, &ws.cub_cap, ctx->stream);
                if(e != hipSuccess) return hip_fail(e, "wp_scan");
            }
            uint64_t tot[3];
            uint32_t stats[4];
            for(int j = 0; j < 3; ++j)
                HIP_TRY(hipMemcpyAsync(&tot[j], ws.d_sz.p + (size_t)j * (n_ent + 1) + n_ent, sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(hipMemcpyAsync(stats, ws.d_small.p, sizeof(stats), hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(hipStreamSynchronize(ctx->stream));
            HIP_TRY(ws.persist.alloc(tot[0] + 64, &a.arena_q));
            HIP_TRY(ws.persist.alloc(tot[2] + 64, &a.arena_path));
            HIP_TRY(ws.d_prep.reserve(tot[1] + 64));
            a.arena_prep = ws.d_prep.p;
            e = launch_wp_materialize(a, ctx->stream);
            if(e != hipSuccess) return hip_fail(e, "wp_materialize");

            // launch order of round 0: long walks first (the few walks across long gaps need bigger path slots: own launches)
            const uint32_t* ext_list = a.list;
            uint32_t n_big = 0, n_mid = 0, n_long_cap = 0;
            bool long_launch = false, long_pending = false;
            if(round == 0) {
                // list_tmp = slot_base + i
                hlist.resize(n_ent);
                for(uint32_t i = 0; i < n_ent; ++i) hlist[i] = (uint32_t)slot_base + i;
                HIP_TRY(hipMemcpyAsync(ws.d_list_tmp.p, hlist.data(), (size_t)n_ent * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
                e = wp_sort_list(ws.d_key.p, ws.d_key_tmp.p, ws.d_list.p, ws.d_list_tmp.p, n_ent, &ws.cub_tmp, &ws.cub_cap, ctx->stream);
                if(e != hipSuccess) return hip_fail(e, "wp_sort_list");
                ext_list = ws.d_list.p;
                n_mid = stats[0]; n_big = stats[1];
            }
            // long_mode 2: the long-gap walks start FIRST, on the side stream with a small share of the wavefront slots (their launch is as
            // long as its longest single walk, not as its walk count), the bulk beside them with the rest; the bulk's failures go to the
            // DP stage while the long walks are still running, theirs in a second call
            const bool long_first = round == 0 && long_mode == 2 && !use_sched && !p.no_dp && n_mid != 0 && n_mid < n_ent;
            WpArgs xl_first = a;
            if(long_first) {
                HIP_TRY(ws.d_items2.reserve(n_mid));
                xl_first.dp_items = ws.d_items2.p; xl_first.n_dp_items = ws.d_small.p + 12; xl_first.dp_items_cap = n_mid;
                HIP_TRY(ws.d_ctr2.reserve(kCtrShards));
                HIP_TRY(hipMemsetAsync(ws.d_ctr2.p, 0, kCtrShards * sizeof(DevCounters), ctx->stream));
                xl_first.ctr = ws.d_ctr2.p;
                if(!ws.ev_side_t0) HIP_TRY(hipEventCreate(&ws.ev_side_t0));
            }
            const int st = timed_launch(ctx, LRSC_K_EXTEND, [&]() -> hipError_t {
                hipError_t e2 = launch_wp_prepare(ctx->fm, a, ctx->stream);
                if(e2 == hipSuccess) e2 = launch_wp_begin(ctx->fm, a, ctx->stream);
                if(e2 != hipSuccess) return e2;
                if(long_first) {
                    e2 = side_begin();
                    if(e2 == hipSuccess) e2 = hipEventRecord(ws.ev_side_t0, ws.side[0]);
                    side_div = long_first_div;
                    reserve_side = true;
                    if(e2 == hipSuccess) e2 = extend_range(xl_first, ext_list, nullptr, n_mid, stats[2], ws.side[0], 1, mid_stride);
                    if(e2 == hipSuccess) e2 = extend_range(a, ext_list + n_mid, nullptr, n_ent - n_mid, std::min(stats[2], kWpPathwSmall), ctx->stream, 0, 1);
                    reserve_side = false;
                    side_div = 2;
                    n_long_cap = n_mid;
                    long_pending = e2 == hipSuccess;
                    return e2;
                }
                if(!use_sched) {
                    if(round != 0) return extend_range(a, a.list, a.reqs, n_ent, std::max(stats[2], 1u), ctx->stream, 0, n_ent <= 16384 ? 64u : n_ent <= 65536 ? 16u : 1u);
                    // the walks across long gaps (the first n_mid of the launch order) run thinly spread over wavefronts: a lane-per-walk
                    // wavefront advances at the pace of its slowest lane, and these are thousands of wide steps long.  They follow the bulk
                    // (both launches are persistent and want every wavefront slot).
                    if(long_mode == 1 && n_mid != 0) {
                        // beside the bulk from the start, each launch with half of the wavefront slots; one DP round for all failures
                        e2 = side_begin();
                        reserve_side = true;
                        if(e2 == hipSuccess) e2 = extend_range(a, ext_list, nullptr, n_mid, stats[2], ws.side[0], 1, mid_stride);
                        if(e2 == hipSuccess) e2 = extend_range(a, ext_list + n_mid, nullptr, n_ent - n_mid, std::min(stats[2], kWpPathwSmall), ctx->stream, 0, 1);
                        reserve_side = false;
                        if(e2 == hipSuccess) e2 = side_join();
                        return e2;
                    }
                    e2 = extend_range(a, ext_list + n_mid, nullptr, n_ent - n_mid, std::min(stats[2], kWpPathwSmall), ctx->stream, 0, 1);
                    if(e2 != hipSuccess || n_mid == 0) return e2;
                    if(p.no_dp) return extend_range(a, ext_list, nullptr, n_mid, stats[2], ctx->stream, 0, mid_stride);
                    // with the DP fallback on they start on a side stream once the bulk is through and share the device with the DP
                    // stage of the bulk's failed walks (own DP item list; half of the wavefront slots)
                    long_launch = true;
                    return e2;
                }
                Pool pools[2];
                e2 = side_begin();
                if(e2 != hipSuccess) return e2;
                if(round == 0) {
                    // the few walks across the longest gaps: one-kernel form on a side stream; the mid class: its own pool on the other
                    e2 = extend_range(a, ext_list, nullptr, n_big, stats[2], ws.side[0], 1, 64);
                    if(e2 == hipSuccess) e2 = pool_setup(pools[0], 0, ext_list + n_mid, n_ent - n_mid, std::max(1u, std::min(stats[2], kWpPathwSmall)), lane_budget, ctx->stream);
                    if(e2 == hipSuccess) e2 = pool_setup(pools[1], 1, ext_list + n_big, n_mid - n_big, std::max(1u, std::min(stats[2], kWpPathwMid)), lane_budget / 4, ws.side[1]);
                } else if(stats[2] > kWpPathwMid)
                    e2 = extend_range(a, a.list, a.reqs, n_ent, stats[2], ctx->stream, 0, 64);
                else {
                    // later rounds: a handful of re-queued walks, one pool (entries that are DP requests or carry a bad geometry end at once)
                    e2 = pool_setup(pools[0], 0, a.list, n_ent, std::max(1u, stats[2]), lane_budget, ctx->stream);
                }
                if(e2 == hipSuccess) e2 = pools_run(a, pools, 2);
                if(e2 == hipSuccess) e2 = side_join();
                return e2;
            });
            if(st != LRSC_OK) return st;
            if(long_launch) {
                long_launch = false;
                n_long_cap = n_mid;
                HIP_TRY(ws.d_items2.reserve(n_mid));
                WpArgs xl = a;
                xl.dp_items = ws.d_items2.p; xl.n_dp_items = ws.d_small.p + 12; xl.dp_items_cap = n_mid;
                // its own statistics counters: the DP stage's timed launches zero and read the ctx's while it runs
                HIP_TRY(ws.d_ctr2.reserve(kCtrShards));
                HIP_TRY(hipMemsetAsync(ws.d_ctr2.p, 0, kCtrShards * sizeof(DevCounters), ctx->stream));
                xl.ctr = ws.d_ctr2.p;
                if(a.prof) xl.prof = a.prof + 16;
                HIP_TRY(side_begin());
                if(!ws.ev_side_t0) HIP_TRY(hipEventCreate(&ws.ev_side_t0));
                HIP_TRY(hipEventRecord(ws.ev_side_t0, ws.side[0]));
                e = extend_range(xl, ext_list, nullptr, n_mid, stats[2], ws.side[0], 1, mid_stride);
                if(e != hipSuccess) return hip_fail(e, "wp_extend (long walks)");
                long_pending = true;
            }

            // ---- the DP stage for every failed walk of this round (and the explicit requests) ---------------------------------
            uint32_t n_items = 0;
            // one DP call over `its` (sorted by slot): requests, DpStage, a persistent copy of the consensus buffer, answers into the slots
            auto dp_call = [&](std::vector<WpDpItem>& its) -> int {
                const uint32_t cnt = (uint32_t)its.size();
                if(cnt == 0) return LRSC_OK;
                reqs.clear();
                reqs.reserve(cnt);
                for(const WpDpItem& it : its) {
                    DpRequest q;
                    std::memset(&q, 0, sizeof(q));
                    q.q_off = it.q;                                                          // absolute device address (base pointer 0)
                    q.lq = it.lq; q.k = it.k;
                    q.coverage = (uint32_t)p.pb_coverage;
                    q.min_overlap = it.lq / 10;                                              // path.length() / 10
                    // identity / min_call_coverage from the two seeds' maxFixedMerFreq (:225-229)
                    const size_t total = (size_t)it.total_freq;
                    double identity = 0.65;
                    size_t min_call_coverage = 15;
                    identity += (total > 50 ? 0.05 : 0);
                    identity += (total > 100 ? 0.05 : 0);
                    min_call_coverage = total > 50 ? total * 0.4 : min_call_coverage;
                    q.min_identity = identity; q.min_call_coverage = (int32_t)min_call_coverage;
                    reqs.push_back(q);
                }
                const int sd = stage.run(ctx, nullptr, reqs);
                if(sd != LRSC_OK) return sd;
                uint8_t* cons_keep = nullptr;
                HIP_TRY(ws.persist.alloc(stage.cons_total + 64, &cons_keep));
                HIP_TRY(hipMemcpyAsync(cons_keep, stage.d_cons.p, stage.cons_total, hipMemcpyDeviceToDevice, ctx->stream));
                HIP_TRY(ws.d_items3.reserve(cnt));
                HIP_TRY(hipMemcpyAsync(ws.d_items3.p, its.data(), (size_t)cnt * sizeof(WpDpItem), hipMemcpyHostToDevice, ctx->stream));
                WpArgs c2 = a;
                c2.dp_reqs = stage.d_reqs.p; c2.dp_msa = stage.d_msa.p; c2.dp_cons = cons_keep; c2.n_dp = cnt;
                hipError_t ec = launch_wp_dp_collect(c2, ws.d_items3.p, ctx->stream);
                if(ec != hipSuccess) return hip_fail(ec, "wp_dp_collect");
                HIP_TRY(hipStreamSynchronize(ctx->stream));                                  // the stage's buffers are reused by the next call
                return LRSC_OK;
            };
            auto fetch_items = [&](DevBuf<WpDpItem>& d_list, const uint32_t* d_count, uint32_t cap, std::vector<WpDpItem>& out) -> int {
                uint32_t cnt = 0;
                HIP_TRY(hipMemcpy(&cnt, d_count, sizeof(uint32_t), hipMemcpyDeviceToHost));
                if(cnt > cap) return fail(LRSC_ERR_LIMIT, "walk-parallel flow: DP item list overflow");
                n_items += cnt;
                out.resize(cnt);
                if(cnt) HIP_TRY(hipMemcpy(out.data(), d_list.p, (size_t)cnt * sizeof(WpDpItem), hipMemcpyDeviceToHost));
                std::sort(out.begin(), out.end(), [](const WpDpItem& x, const WpDpItem& y) { return x.slot < y.slot; });
                return LRSC_OK;
            };
            {
                int sd = fetch_items(ws.d_items, a.n_dp_items, n_ent, items);
                if(sd != LRSC_OK) return sd;
                if(!long_pending) sd = dp_call(items);
                else {
                    // The long-gap walks are running on the side stream.  The bulk's failed walks go to the DP stage in two calls: the
                    // first 60 % now; the rest together with the long walks' failures, which are in by then -- their few, long
                    // alignments and pile-ups (latency-bound on their own) then overlap the second call's bulk instead of trailing it.
                    const size_t cut = items.size() * dp_split_pct / 100;
                    std::vector<WpDpItem> second(items.begin() + (ptrdiff_t)cut, items.end());
                    items.resize(cut);
                    sd = dp_call(items);
                    if(sd != LRSC_OK) return sd;
                    if(!ws.ev_side_t1) HIP_TRY(hipEventCreate(&ws.ev_side_t1));
                    HIP_TRY(hipEventRecord(ws.ev_side_t1, ws.side[0]));
                    HIP_TRY(hipEventSynchronize(ws.ev_side_t1));
                    float ms = 0.f;
                    HIP_TRY(hipEventElapsedTime(&ms, ws.ev_side_t0, ws.ev_side_t1));
                    ctx->stats[LRSC_K_EXTEND].total_ms += ms;             // overlaps the DP stage: the stage times then add up to more than the wall time
                    {
                        std::vector<DevCounters> shards(kCtrShards);
                        HIP_TRY(hipMemcpy(shards.data(), ws.d_ctr2.p, kCtrShards * sizeof(DevCounters), hipMemcpyDeviceToHost));
                        for(const DevCounters& dcn : shards) {
                            ctx->stats[LRSC_K_EXTEND].rank_queries += dcn.rank_queries;
                            ctx->stats[LRSC_K_EXTEND].block_loads += dcn.block_loads;
                            ctx->stats[LRSC_K_EXTEND].table_loads += dcn.table_loads;
                        }
                    }
                    long_pending = false;
                    std::vector<WpDpItem> longs;
                    sd = fetch_items(ws.d_items2, ws.d_small.p + 12, n_long_cap, longs);
                    if(sd != LRSC_OK) return sd;
                    second.insert(second.end(), longs.begin(), longs.end());
                    std::sort(second.begin(), second.end(), [](const WpDpItem& x, const WpDpItem& y) { return x.slot < y.slot; });
                    sd = dp_call(second);
                }
                if(sd != LRSC_OK) return sd;
            }
            HIP_TRY(hipMemsetAsync(a.n_req_out, 0, sizeof(uint32_t), ctx->stream));
            e = launch_wp_stitch(a, ctx->stream);
            if(e != hipSuccess) return hip_fail(e, "wp_stitch");
            ++rounds_total;
            if(verbose && round == 0 && std::getenv("LRSC_WP_DUMP")) {
                // profiling aid: the hardest walks of the range (a walk is a chain of dependent steps: they bound the launch from below)
                std::vector<WpSlot> hs(n_range);
                HIP_TRY(hipStreamSynchronize(ctx->stream));
                HIP_TRY(hipMemcpy(hs.data(), ws.d_slots.p + slot_base, (size_t)n_range * sizeof(WpSlot), hipMemcpyDeviceToHost));
                std::vector<uint32_t> ord(n_range);
                for(uint32_t i = 0; i < n_range; ++i) ord[i] = i;
                std::sort(ord.begin(), ord.end(), [&](uint32_t x, uint32_t y) { return hs[x].leaf_steps > hs[y].leaf_steps; });
                unsigned long long tot = 0, tot_steps = 0;
                for(const WpSlot& q : hs) { tot += q.leaf_steps; tot_steps += q.steps; }
                std::fprintf(stderr, "[lrsc] wp walks of the range: %u, %llu steps, %llu leaf-steps; hardest (gap, k, steps, leaf-steps, code):", n_range, tot_steps, tot);
                for(uint32_t i = 0; i < std::min<uint32_t>(n_range, 12); ++i)
                    std::fprintf(stderr, " (%u,%u,%u,%u,%d)", hs[ord[i]].gap, (unsigned)hs[ord[i]].k, hs[ord[i]].steps, hs[ord[i]].leaf_steps, hs[ord[i]].code);
                {
                    unsigned long long hist[34] = {0}, wsteps[34] = {0};
                    for(const WpSlot& q : hs) { const unsigned m = std::min<unsigned>(q.max_front, 33); hist[m]++; wsteps[m] += q.leaf_steps; }
                    std::fprintf(stderr, "; walks (and their leaf-steps in %%) by widest frontier:");
                    unsigned long long cw = 0, cs = 0;
                    for(unsigned m = 1; m < 34; ++m) { cw += hist[m]; cs += wsteps[m]; if(hist[m]) std::fprintf(stderr, " %u:%.2f%%(%.1f%%)", m, 100.0 * hist[m] / n_range, 100.0 * wsteps[m] / std::max(tot, 1ull)); }
                }
                const uint32_t qs[] = {n_range / 2, n_range / 10, n_range / 100, n_range / 1000, n_range / 10000};
                std::fprintf(stderr, "; leaf-steps at the median / top 10%% / 1%% / 0.1%% / 0.01%%: %u %u %u %u %u\n", hs[ord[qs[0]]].leaf_steps, hs[ord[qs[1]]].leaf_steps,
                             hs[ord[qs[2]]].leaf_steps, hs[ord[qs[3]]].leaf_steps, hs[ord[qs[4]]].leaf_steps);
            }
            if(verbose)
                std::fprintf(stderr, "[lrsc] wp reads [%u, %u) round %u: %u entries, %u DP requests (%llu strings), arenas q %.1f MB prep %.1f MB path %.1f MB, %llu schedule rounds so far\n",
                             r0, r1, round, n_ent, n_items, (unsigned long long)stage.n_strings, tot[0] / 1048576.0, tot[1] / 1048576.0, tot[2] / 1048576.0,
                             (unsigned long long)sched_rounds);
            if(round > 100000) return fail(LRSC_ERR_LIMIT, "walk-parallel flow: too many rounds");
        }
        r0 = r1;
    }
    (void)rounds_total;
    if(a.prof) {
        unsigned long long prs[32];
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        for(int i = 0; i < 2; ++i) HIP_TRY(hipStreamSynchronize(ws.side[i]));
        HIP_TRY(hipMemcpy(prs, ws.d_prof.p, sizeof(prs), hipMemcpyDeviceToHost));
        for(int part = 0; part < 2; ++part) {
            const unsigned long long* pr = prs + 16 * part;
            if(pr[10] == 0) continue;
            const double all = (double)pr[10], st = (double)std::max<unsigned long long>(pr[11], 1);
            std::fprintf(stderr, "[lrsc] wp extension kernel (%s), lane wall ticks: %.3g total, %.0f per step over %.3g steps; extendLeaves %.1f%% (refine %.1f%%, attempToExtend %.1f%% of which "
                                 "getFMIndexExtensions %.1f%%), PrunedBySeedSupport %.1f%%, materialise+commit %.1f%%, isTerminated %.1f%%, refill %.1f%%, finish %.1f%%; single-leaf fast steps %.1f%% of the steps in %.1f%% of the ticks\n",
                         part ? "long-gap walks, one per wavefront" : "bulk and later rounds",
                         all, all / st, st, 100 * pr[0] / all, 100 * pr[1] / all, 100 * pr[2] / all, 100 * pr[3] / all, 100 * pr[4] / all, 100 * pr[5] / all, 100 * pr[6] / all,
                         100 * pr[8] / all, 100 * pr[9] / all, 100 * pr[12] / st, 100 * pr[7] / all);
        }
    }
    if(a.walk_log) b->walk_log_done = true;

    // ---- results ---------------------------------------------------------------------------------------------------------------------
    std::vector<WpRead> ro(n);
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    HIP_TRY(hipMemcpy(ro.data(), ws.d_reads.p, (size_t)n * sizeof(WpRead), hipMemcpyDeviceToHost));
    std::vector<uint32_t> pieces(std::max<uint64_t>(piece_total, 1));
    if(piece_total) HIP_TRY(hipMemcpy(pieces.data(), cs.d_pieces.p, (size_t)piece_total * sizeof(uint32_t), hipMemcpyDeviceToHost));
    std::vector<uint64_t> dst_off(n + 1, 0);
    uint64_t n_pieces = 0;
    for(uint32_t r = 0; r < n; ++r) {
        const WpRead& o = ro[r];
        int status = skipped[r];
        if(o.error == LRSC_WALK_ERR_GEOMETRY) status = LRSC_READ_GEOMETRY;
        else if(o.error == LRSC_WALK_ERR_CODE) status = LRSC_READ_INTERNAL;
        else if(o.error == LRSC_WALK_ERR_DP) status = LRSC_READ_DP_LIMIT;
        else if(o.error == LRSC_WALK_ERR_OUTPUT) status = LRSC_READ_OUTPUT_LIMIT;
        else if(o.error != 0) status = LRSC_READ_FRONTIER_LIMIT;
        lrsc_read_result& R = res[r];
        if(status != LRSC_READ_OK) {
            // this read alone could not be corrected: it comes back as "not merged" (-> discard.fa) with its status
            std::memset(&R, 0, sizeof(R));
            R.piece_first = n_pieces;
            R.status = status;
            dst_off[r + 1] = dst_off[r];
            continue;
        }
        R.merge = (int32_t)o.merge; R.n_pieces = o.n_pieces; R.piece_first = n_pieces;
        R.total_reads_len = (int64_t)(off[r + 1] - off[r]); R.corrected_len = o.c[1]; R.total_seed_num = seed_count[r]; R.total_walk_num = o.c[3];
        R.high_error_num = o.c[4]; R.exceed_depth_num = o.c[5]; R.exceed_leave_num = o.c[6]; R.fm_num = o.c[7];
        R.dp_num = o.c[8]; R.seed_dis = o.c[9];
        R.status = LRSC_READ_OK; R.pad = 0;
        dst_off[r + 1] = dst_off[r] + o.out_len;
        for(uint32_t j = 0; j < o.n_pieces; ++j) {
            if(piece_off && n_pieces < piece_cap) piece_off[n_pieces] = dst_off[r] + pieces[work[r].piece_off + j];
            ++n_pieces;
        }
    }
    const uint64_t used = dst_off[n];
    if(piece_off && n_pieces < piece_cap) piece_off[n_pieces] = used;
    *n_pieces_out = n_pieces;
    *out_used = used;
    if(!out || !piece_off || used > out_cap || n_pieces + 1 > piece_cap) return fail(LRSC_ERR_CAPACITY, "output buffers too small");
    if(used) {
        HIP_TRY(cs.d_dst_off.reserve(n + 1));
        HIP_TRY(cs.d_dst.reserve(used));
        HIP_TRY(hipMemcpyAsync(cs.d_dst_off.p, dst_off.data(), (size_t)(n + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
        e = launch_wp_gather(a, cs.d_dst_off.p, cs.d_dst.p, ctx->stream);
        if(e != hipSuccess) return hip_fail(e, "wp_gather");
        HIP_TRY(hipMemcpyAsync(out, cs.d_dst.p, used, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
    }
    return LRSC_OK;
}

// ---------------------------------------------------------------------------------------
// the whole per-read path on the device
// ---------------------------------------------------------------------------------------
// PacBioSelfCorrectionProcess::process for a resident batch: seeds (if not found yet), then the walk-parallel flow
// (batch_correct_wp above: wp.hip) and a gather of the corrected strings.  The host only sizes buffers and copies results.
extern "C" int lrsc_batch_correct(lrsc_ctx* ctx, lrsc_batch* b, lrsc_read_result* res, uint64_t* piece_off, uint64_t piece_cap,
                                  char* out, uint64_t out_cap, uint64_t* n_pieces_out, uint64_t* out_used)
{
    if(!ctx || !b || b->ctx != ctx || !res || !n_pieces_out || !out_used) return fail(LRSC_ERR_ARG, "null / foreign batch");
    *n_pieces_out = 0; *out_used = 0;
    const lrsc_params& p = ctx->params;
    if(p.max_leaves < 1 || p.max_leaves > 32) return fail(LRSC_ERR_UNSUPPORTED, "max_leaves must be 1..32");
    if(p.idmer_len < 5 || p.idmer_len > 16) return fail(LRSC_ERR_UNSUPPORTED, "idmer_len must be 5..16");
    if(p.min_kmer_len < p.idmer_len || p.min_kmer_len > 62) return fail(LRSC_ERR_UNSUPPORTED, "min_kmer_len out of range");
    if(p.next_target < 1) return fail(LRSC_ERR_ARG, "next_target must be >= 1");
    const uint32_t n = b->n_reads;
    if(n == 0) return LRSC_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    if(!b->seeds_done) {
        const int st = lrsc_batch_find_seeds(ctx, b);
        if(st != LRSC_OK) return st;
    }
    return batch_correct_wp(ctx, b, res, piece_off, piece_cap, out, out_cap, n_pieces_out, out_used);
}

// ---------------------------------------------------------------------------------------
// DP/MSA fallback
// ---------------------------------------------------------------------------------------
extern "C" int lrsc_dp_align(lrsc_ctx* ctx, const char* seq, uint64_t seq_len, const lrsc_dp_job* jobs, uint32_t n, int band_width,
                             int match_score, int gap_penalty, int mismatch_penalty, lrsc_dp_result* results, char* cigar_arena,
                             uint64_t arena_cap, uint64_t* arena_used)
{
    if(!ctx || (!jobs && n) || (!results && n) || !arena_used || (!seq && seq_len)) return fail(LRSC_ERR_ARG, "null");
    *arena_used = 0;
    if(n == 0) return LRSC_OK;
    if(band_width < 2 || (band_width / 2) * 2 + 1 > (int)kDpMaxBand) return fail(LRSC_ERR_UNSUPPORTED, "band_width must be 2..254");
    if(gap_penalty > 0) return fail(LRSC_ERR_ARG, "gap_penalty must be <= 0");
    HIP_TRY(hipSetDevice(ctx->device));
    std::vector<uint8_t> codes;
    int st = encode_acgt(seq, seq_len, codes);
    if(st != LRSC_OK) return st;
    std::vector<DpJob> dj(n);
    uint64_t ops_total = 0;
    uint32_t max1 = 1, max2 = 1;
    for(uint32_t i = 0; i < n; ++i) {
        const lrsc_dp_job& j = jobs[i];
        if(j.s1_off + j.s1_len > seq_len || j.s2_off + j.s2_len > seq_len) return fail(LRSC_ERR_ARG, "dp job: sequence out of range");
        if(j.s1_len > kDpMaxSeq || j.s2_len > kDpMaxSeq) return fail(LRSC_ERR_UNSUPPORTED, "dp job: sequence too long");
        DpJob& d = dj[i];
        d.s1_off = j.s1_off; d.s2_off = j.s2_off; d.s1_len = j.s1_len; d.s2_len = j.s2_len; d.start1 = j.start1; d.start2 = j.start2;
        d.mode = 0; d.req = 0; d.ops_off = ops_total;
        ops_total += (uint64_t)j.s1_len + j.s2_len + 1;
        max1 = std::max(max1, j.s1_len); max2 = std::max(max2, j.s2_len);
    }
    // sequences beyond the 64 KB LDS stage: the same kernel with its staging in a global slice per wavefront (fewer wavefronts)
    const bool global_stage = dp_align_stage_bytes(max1, max2) > kDpAlignLdsCap;
    const uint64_t stage_stride = (dp_align_stage_bytes(max1, max2) + 255) & ~255ull;
    const uint32_t n_waves = std::min<uint32_t>(global_stage ? 1024u : dp_wave_count(ctx), n);
    DevBuf<uint8_t> d_codes, d_ops, d_trace, d_stage;
    DevBuf<DpJob> d_jobs;
    DevBuf<DpAlignOut> d_out;
    DpAlignArgs a{};
    a.trace_stride = (uint64_t)(max1 + 17) * kDpTraceStride;
    HIP_TRY(d_codes.reserve(std::max<uint64_t>(seq_len, 1)));
    HIP_TRY(d_ops.reserve(ops_total));
    HIP_TRY(d_trace.reserve(a.trace_stride * n_waves));
    HIP_TRY(d_jobs.reserve(n));
    HIP_TRY(d_out.reserve(n));
    HIP_TRY(hipMemcpyAsync(d_codes.p, codes.data(), seq_len, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(d_jobs.p, dj.data(), (size_t)n * sizeof(DpJob), hipMemcpyHostToDevice, ctx->stream));
    a.codes = d_codes.p; a.strings = d_codes.p; a.jobs = d_jobs.p; a.n_jobs = n; a.band_width = (uint32_t)band_width;
    a.match_score = match_score; a.gap_penalty = gap_penalty; a.mismatch_penalty = mismatch_penalty;
    a.ops = d_ops.p; a.out = d_out.p; a.trace = d_trace.p; a.max_s1 = max1; a.max_s2 = max2;
    a.lds_cap = kDpAlignLdsCap;
    if(global_stage) {
        HIP_TRY(d_stage.reserve(stage_stride * n_waves));
        a.seq_ws = d_stage.p; a.seq_ws_stride = stage_stride; a.only_long = 0;
    }
    st = timed_launch(ctx, LRSC_K_DP, [&]() { return launch_dp_align(a, n_waves, ctx->stream); });
    if(st != LRSC_OK) return st;
    std::vector<DpAlignOut> out(n);
    std::vector<uint8_t> ops(ops_total);
    HIP_TRY(hipMemcpy(out.data(), d_out.p, (size_t)n * sizeof(DpAlignOut), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(ops.data(), d_ops.p, ops_total, hipMemcpyDeviceToHost));
    uint64_t used = 0;
    for(uint32_t i = 0; i < n; ++i) {
        const DpAlignOut& o = out[i];
        if(o.n_ops == 0xFFFFFFFFu) return fail(LRSC_ERR_DEVICE, "dp_align: traceback left the band");
        lrsc_dp_result& r = results[i];
        r.match0_start = o.m0s; r.match0_end = o.m0e; r.match1_start = o.m1s; r.match1_end = o.m1e;
        r.score = o.score; r.edit_distance = o.edit_distance; r.total_columns = o.total_columns;
        r.cigar_len = o.n_ops; r.cigar_off = used;
        if(cigar_arena && used + o.n_ops <= arena_cap)
            for(uint32_t t = 0; t < o.n_ops; ++t) cigar_arena[used + t] = (char)ops[dj[i].ops_off + o.n_ops - 1 - t];
        used += o.n_ops;
    }
    *arena_used = used;
    if(used > arena_cap || (!cigar_arena && used)) return fail(LRSC_ERR_CAPACITY, "cigar arena too small");
    return LRSC_OK;
}

extern "C" int lrsc_dp_consensus(lrsc_ctx* ctx, const char* seq, uint64_t seq_len, const lrsc_msa_query* queries, uint32_t n,
                                 lrsc_msa_result* results, char* arena, uint64_t arena_cap, uint64_t* arena_used)
{
    if(!ctx || (!queries && n) || (!results && n) || !arena_used || (!seq && seq_len)) return fail(LRSC_ERR_ARG, "null");
    *arena_used = 0;
    if(n == 0) return LRSC_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    std::vector<uint8_t> codes;
    int st = encode_acgt(seq, seq_len, codes);
    if(st != LRSC_OK) return st;
    std::vector<DpRequest> reqs(n);
    for(uint32_t i = 0; i < n; ++i) {
        const lrsc_msa_query& q = queries[i];
        if(q.seq_off + q.len > seq_len) return fail(LRSC_ERR_ARG, "msa query out of range");
        DpRequest& r = reqs[i];
        std::memset(&r, 0, sizeof(r));
        r.q_off = q.seq_off; r.lq = q.len; r.k = q.kmer_len; r.min_overlap = q.min_overlap; r.min_call_coverage = q.min_call_coverage;
        r.min_identity = q.min_identity; r.coverage = (uint32_t)ctx->params.pb_coverage;
    }
    DevBuf<uint8_t> d_codes;
    HIP_TRY(d_codes.reserve(std::max<uint64_t>(seq_len, 1)));
    HIP_TRY(hipMemcpyAsync(d_codes.p, codes.data(), seq_len, hipMemcpyHostToDevice, ctx->stream));
    DpStage stage;
    st = stage.run(ctx, d_codes.p, reqs);
    if(st != LRSC_OK) return st;
    std::vector<DpMsaOut> mo(n);
    std::vector<uint8_t> cons(stage.cons_total);
    HIP_TRY(hipMemcpy(mo.data(), stage.d_msa.p, (size_t)n * sizeof(DpMsaOut), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(cons.data(), stage.d_cons.p, stage.cons_total, hipMemcpyDeviceToHost));
    uint64_t used = 0;
    for(uint32_t i = 0; i < n; ++i) {
        if(mo[i].error) return fail(LRSC_ERR_LIMIT, "msa: column capacity exceeded");
        results[i].n_rows = mo[i].n_rows; results[i].n_retrieved = reqs[i].n_str; results[i].cons_len = mo[i].cons_len; results[i].rows_by_step_walk = mo[i].pad;
        results[i].cons_off = used;
        if(arena && used + mo[i].cons_len <= arena_cap)
            for(uint32_t t = 0; t < mo[i].cons_len; ++t) arena[used + t] = "ACGT"[cons[reqs[i].cons_off + t] & 3u];
        used += mo[i].cons_len;
    }
    *arena_used = used;
    if(used > arena_cap || (!arena && used)) return fail(LRSC_ERR_CAPACITY, "consensus arena too small");
    return LRSC_OK;
}
