// correct_layout.h -- host-side sizing of one read's correction workspace and output slot (shared by capi.cpp and the
// CPU emulation harness under tests/host_emul, so that both drive the kernels' state machine over the same layout).
#pragma once
#include <algorithm>
#include <cstring>

#include "correct_dev.h"
#include "ws_layout.h"
#include "introsort_emul.h"

namespace lrsc {

inline size_t layout_align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// Fills w (offsets relative to the read's workspace base) and returns the workspace bytes; 0 with *err set on a limit.
// rlen = read length, ns = its seed count (>= 2), plan = bounds of its walks, psz = 4 / 8 (narrow / wide positions),
// lbytes = sizeof(Leaf<P>).
inline size_t layout_read_work(ReadWork& w, uint64_t rlen, uint32_t ns, const ReadPlan& plan, bool no_dp, bool split, uint32_t idmer_len,
                               size_t psz, size_t lbytes, const char** err, int* code = nullptr)
{
    *err = nullptr;
    int code_local = 0;
    if(!code) code = &code_local;
    *code = 0;
    // every walk appends at most maxLength + 1 + |target| - initk characters, walks <= seeds, gaps sum to <= |read|
    // (a DP consensus can be longer than its query by the insertion columns it keeps: budget 2x the raw segment)
    const uint64_t cap = rlen + (uint64_t)((no_dp ? 1.2 : 2.0) * (double)rlen) + (uint64_t)ns * (2 * kMaxInitK + 16 + (no_dp ? 0 : 128)) + 64;
    if(cap >= (1ull << 32)) { *err = "read too long"; *code = 2; return 0; }
    w.out_cap = (uint32_t)cap;
    w.piece_cap = split ? ns : 1;
    w.lq_max = plan.lq_max;
    if(w.lq_max >= 65535) { *err = "walk: query longer than 65534 bases"; *code = 1; return 0; }
    const double maxLength = (1.2 * ((double)plan.gap_max + 10)) + (double)(2 * (uint64_t)kMaxInitK);
    w.pathw = (uint32_t)(((uint64_t)maxLength + 4 + 15) / 16 + 1);
    // fixed-size regions first (their offsets are compile-time constants for the state-machine kernel), then the ones that
    // scale with the longest query / path of the read: ws_layout.h holds the one set of formulas both sides use
    const WsVar v = ws_var_offsets((uint32_t)lbytes, (uint32_t)psz, w.lq_max, idmer_len, w.pathw);
    w.o_leaves = ws_fixed_leaves(); w.o_rings = ws_fixed_rings((uint32_t)lbytes); w.o_results = ws_fixed_results((uint32_t)lbytes);
    w.o_head9 = ws_fixed_head9((uint32_t)lbytes); w.o_head5 = ws_fixed_head5((uint32_t)lbytes);
    w.o_item9f = v.item9f; w.o_item9r = v.item9r; w.o_term = v.term; w.o_paths = v.paths; w.o_best = v.best;
    w.o_next9f = v.next9f; w.o_next9r = v.next9r; w.o_next5 = v.next5; w.o_flags5 = v.flags5; w.o_query = v.query; w.o_dpq = v.dpq;
    size_t o = v.total;
    if(o >= (1ull << 32)) { *err = "read workspace too large"; *code = 2; return 0; }
    return o;
}

// correct_plan_kernel's arithmetic for one read (bounds of every walk it can be asked for)
inline ReadPlan plan_read(const int32_t* seeds, uint32_t n_seeds, int next_target)
{
    ReadPlan p{0, 0};
    for(uint32_t it = 1; it < n_seeds; ++it) {
        const int s_end = seeds[(uint64_t)(it - 1) * 8] + seeds[(uint64_t)(it - 1) * 8 + 1] - 1;
        for(int next = 0; next < next_target && it + (uint32_t)next < n_seeds; ++next) {
            const int32_t* T = seeds + (uint64_t)(it + (uint32_t)next) * 8;
            const int gap = T[0] - s_end - 1;
            if(gap < 0) continue;
            if((uint32_t)gap > p.gap_max) p.gap_max = (uint32_t)gap;
            const uint32_t lq = kMaxInitK + (uint32_t)gap + (uint32_t)T[1];
            if(lq > p.lq_max) p.lq_max = lq;
        }
    }
    return p;
}

} // namespace lrsc
