// ws_layout.h -- byte offsets of the regions of one read's correction workspace, as functions of the few per-read sizes.
// One set of formulas for the host (correct_layout.h fills ReadWork from them) and for the state-machine kernel
// (walk_sm.h recomputes an offset where it needs it instead of carrying sixteen of them in registers).
#pragma once
#include <stdint.h>

#include "extend.h"

#ifdef __HIPCC__
#define LRSC_WS_HD __host__ __device__ inline
#else
#define LRSC_WS_HD inline
#endif

namespace lrsc {

LRSC_WS_HD uint32_t ws_al16(uint32_t x) { return (x + 15u) & ~15u; }
// fixed-size regions: leaf frontier (32 + kMaxChildren), error-history rings, result records, 9-mer / 5-mer chain heads
LRSC_WS_HD constexpr uint32_t ws_fixed_leaves() { return 0; }
LRSC_WS_HD uint32_t ws_fixed_rings(uint32_t lbytes) { return ws_al16((32u + kMaxChildren) * lbytes); }
LRSC_WS_HD uint32_t ws_fixed_results(uint32_t lbytes) { return ws_fixed_rings(lbytes) + 32u * 100u * 8u; }
LRSC_WS_HD uint32_t ws_fixed_head9(uint32_t lbytes) { return ws_fixed_results(lbytes) + kMaxResults * (uint32_t)sizeof(WalkResultRec); }
LRSC_WS_HD uint32_t ws_fixed_head5(uint32_t lbytes) { return ws_fixed_head9(lbytes) + 512u * 2u; }
LRSC_WS_HD uint32_t ws_var_base(uint32_t lbytes) { return ws_fixed_head5(lbytes) + 1024u * 2u; }

struct WsVar { uint32_t item9f, item9r, term, paths, best, next9f, next9r, next5, flags5, query, dpq, total; };
// lq_max = longest m_query of the read, pathw = 2-bit path words per slot
LRSC_WS_HD WsVar ws_var_offsets(uint32_t lbytes, uint32_t psz, uint32_t lq_max, uint32_t idmer_len, uint32_t pathw)
{
    const uint32_t lq = lq_max > 16u ? lq_max : 16u;
    const uint32_t n9 = lq - idmer_len + 1u, n5 = lq - 5u + 1u;
    WsVar v;
    uint32_t o = ws_var_base(lbytes);
    v.item9f = o; o += n9 * 16u;                       // SortItem
    v.item9r = o; o += n9 * 16u;
    v.term = o;   o = ws_al16(o + lq * 4u * psz);      // >= |target| - minOverlap + 1 entries
    v.paths = o;  o += (32u + kMaxResults) * pathw * 4u;
    v.best = o;   o += pathw * 4u;
    v.next9f = o; o += n9 * 2u;
    v.next9r = o; o += n9 * 2u;
    v.next5 = o;  o += n5 * 2u;
    v.flags5 = o; o += n5;
    v.query = o;  o += lq;
    v.dpq = o;    o += lq;
    v.total = (o + 63u) & ~63u;
    return v;
}

} // namespace lrsc
