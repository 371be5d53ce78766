// correct_dev.hip -- the persistent correction kernel: ONE LANE PER READ runs that read's whole chain of
// seed-to-seed walks back to back and stitches the corrected string on the device
// (PacBioSelfCorrectionProcess::initCorrect / correctByFMExtension, PacBio/PacBioSelfCorrectionProcess.cpp:56-206,
//  --nodp flavour: a failed walk copies the raw segment or, with --split, starts a new piece).
//
// Why per read and not per walk: the walks of a read are a dependent chain (the next source k-mer is the
// tail of what was just appended), so a per-walk launch has to be repeated once per chain position (~90
// rounds for 10 kb reads) and every round lasts as long as its longest walk.  Every read, however, has
// about read-length extension steps in total, so lanes that each own a read finish together.
#include <hip/hip_runtime.h>

#include "correct_dev.h"
#include "walk_device.h"

namespace lrsc {

template <bool WIDE, int OCC>
__global__ __launch_bounds__(64, OCC) void correct_reads_kernel(FmIndexDev fm, CorrectArgs a)
{
    using P = typename Lay<WIDE>::pos_t;
    __shared__ __attribute__((aligned(16))) uint32_t mtab[MaskTabSize<WIDE>::value];
    init_mask_table<WIDE>(mtab);
    const uint32_t stride = 64u / a.reads_per_wave;
    const uint32_t slot0 = blockIdx.x * a.reads_per_wave + threadIdx.x / stride;
    const bool owner = (threadIdx.x % stride) == 0;
    uint32_t n_rank = 0, n_blk = 0;
    // a.queue set: fewer lanes than reads, every lane pulls the next read of the launch when its current one is done,
    // parked or out of budget (the lanes of a wavefront then stay busy until the queue is empty)
    for(uint32_t slot = slot0, first = 1; owner; first = 0) {
        if(a.queue) slot = atomicAdd(a.queue, 1u);
        else if(!first) break;
        if(slot >= a.n_reads) break;
        const uint32_t r = a.order ? a.order[slot] : slot;
        const ReadWork rw = a.work[r];
        ReadOut& R = a.out[r];
        const uint64_t rs = a.read_off[r];
        const uint32_t rlen = (uint32_t)(a.read_off[r + 1] - rs);
        const uint8_t* read = a.codes + rs;
        const uint32_t n_seeds = a.seed_count[r];
        const int32_t* seeds = a.seeds + seed_slab(rs, r, a.min_k) * kSeedInts;
        uint8_t* out = a.out_codes + rw.out_off;
        uint8_t* wl = a.walk_log ? a.walk_log + seed_slab(rs, r, a.min_k) : nullptr;
        uint32_t* piece_start = a.piece_start + rw.piece_off;

        const bool resume = a.resume != 0;                                       // the read was parked waiting for its DP result
        int64_t correctedLen = 0, totalWalkNum = 0, highErrorNum = 0, exceedDepthNum = 0, exceedLeaveNum = 0, FMNum = 0, DPNum = 0, seedDis = 0;
        uint32_t out_len = 0, n_pieces = 0;
        int error = 0;
        uint32_t state = kReadDone;
        uint64_t cyc_prep = 0, cyc_stitch = 0;
        if(resume) {
            correctedLen = R.c[1]; totalWalkNum = R.c[3]; highErrorNum = R.c[4]; exceedDepthNum = R.c[5]; exceedLeaveNum = R.c[6];
            FMNum = R.c[7]; DPNum = R.c[8]; seedDis = R.c[9];
            out_len = R.out_len; n_pieces = R.n_pieces;
            cyc_prep = R.cyc[0];
        } else { R.cyc[0] = 0; R.cyc[1] = 0; R.cyc[2] = 0; R.cyc[3] = 0; R.steps = 0; for(int k = 0; k < 8; ++k) R.cyc_step[k] = 0; }

        if(n_seeds >= 2 && rw.lq_max != 0 && !(resume && R.state == kReadDone)) {      // lq_max == 0: the host skipped this read (capacity)
            uint8_t* ws = a.workspace + rw.ws_off;
            Walk<WIDE> W;
            W.sF = strand_consts<P>(fm.strand[LRSC_RBWT]);
            W.sR = strand_consts<P>(fm.strand[LRSC_BWT]);
            W.fm = &fm;
            W.mtab = mtab;
            W.seedSize = a.seed_size; W.minOverlap = a.min_overlap; W.maxLeaves = a.max_leaves;
            W.PBcoverage = a.pb_coverage; W.PacBioErrorRate = a.pacbio_error_rate; W.errorRate = 0.25; W.localK = 100;
            W.freqsOfKmerSize = a.freqs_of_kmer_size;
            W.it9f = reinterpret_cast<SortItem*>(ws + rw.o_item9f);
            W.it9r = reinterpret_cast<SortItem*>(ws + rw.o_item9r);
            W.next9f = reinterpret_cast<uint16_t*>(ws + rw.o_next9f);
            W.next9r = reinterpret_cast<uint16_t*>(ws + rw.o_next9r);
            W.head9f = reinterpret_cast<uint16_t*>(ws + rw.o_head9);
            W.head9r = W.head9f + 256;
            W.head5 = reinterpret_cast<uint16_t*>(ws + rw.o_head5);
            W.next5 = reinterpret_cast<uint16_t*>(ws + rw.o_next5);
            uint8_t* flags5 = ws + rw.o_flags5;
            W.flags5 = flags5;
            P* term = reinterpret_cast<P*>(ws + rw.o_term);
            W.term = term;
            W.cur = reinterpret_cast<Leaf<P>*>(ws + rw.o_leaves);
            W.nxt = W.cur + 32;
            W.leaf_small = W.cur;
            W.rings = reinterpret_cast<double*>(ws + rw.o_rings);
            W.paths = reinterpret_cast<uint32_t*>(ws + rw.o_paths);
            W.pathw = rw.pathw;
            W.rpaths = W.paths + (uint64_t)32 * rw.pathw;
            W.results = reinterpret_cast<WalkResultRec*>(ws + rw.o_results);
            uint8_t* q = ws + rw.o_query;
            uint32_t* best = reinterpret_cast<uint32_t*>(ws + rw.o_best);
            W.q = q;
            W.n_rank = 0; W.n_blk = 0; W.steps = R.steps; W.error = 0; W.cyc_setup = R.cyc[1]; W.cyc_loop = R.cyc[2];
            W.profile = a.profile != 0;
            W.prof = a.profile ? R.cyc_step : nullptr;
            uint8_t* dpq = ws + rw.o_dpq;                                          // the parked DP query lives here between launches

            // source = pieceVec.back(): the SeedFeature fields the loop reads (SeedFeature.h:22-45)
            int S_seedLen, S_end, S_endBest, S_maxFixed;
            bool S_isRepeat;
            uint32_t it;
            if(!resume) {
                // pieceVec.push_back(seedVec[0])
                piece_start[n_pieces++] = 0;
                for(int t = 0; t < seeds[1]; ++t) out[out_len++] = read[seeds[0] + t];
                S_seedLen = seeds[1]; S_end = seeds[0] + seeds[1] - 1; S_endBest = seeds[5]; S_maxFixed = seeds[2];
                S_isRepeat = (seeds[3] & 1) != 0;
                it = 1;
            } else if(R.state == kReadYield) {
                S_seedLen = R.s_seed_len; S_end = R.s_end; S_endBest = R.s_end_best; S_maxFixed = R.s_max_fixed; S_isRepeat = R.s_is_repeat != 0;
                it = R.it;
            } else {
                S_seedLen = R.s_seed_len; S_end = R.s_end; S_endBest = R.s_end_best; S_maxFixed = R.s_max_fixed; S_isRepeat = R.s_is_repeat != 0;
                it = R.it;
                // correctByMSAlignment's tail (:237-244) with the DP stage's answer for target = *iterTarget
                const uint32_t di = a.dp_index[r];
                const DpMsaOut m = a.dp_msa[di];
                const int32_t* T0 = seeds + (uint64_t)it * kSeedInts;
                if(m.error) error = LRSC_WALK_ERR_DP;
                else if(m.n_rows > 3) {
                    const uint8_t* cons = a.dp_cons + a.dp_reqs[di].cons_off;
                    if(m.cons_len < R.dp_k) error = LRSC_WALK_ERR_DP;                // out.erase(0, k) would throw in the reference
                    else {
                        const uint32_t appended = m.cons_len - R.dp_k;
                        if(out_len + appended > rw.out_cap) error = LRSC_WALK_ERR_OUTPUT;
                        else {
                            for(uint32_t j = 0; j < appended; ++j) out[out_len + j] = cons[R.dp_k + j];
                            out_len += appended;
                            correctedLen += appended;
                            seedDis += T0[0] - S_end - 1;
                            DPNum++;
                            S_seedLen += (int)appended;
                        }
                    }
                } else if(a.split) {
                    if(wl) wl[it] |= 0x10;
                    if(out_len + (uint32_t)T0[1] > rw.out_cap) error = LRSC_WALK_ERR_OUTPUT;
                    else {
                        piece_start[n_pieces++] = out_len;
                        for(int t = 0; t < T0[1]; ++t) out[out_len++] = read[T0[0] + t];
                        S_seedLen = T0[1];
                        correctedLen += T0[1];
                    }
                } else {
                    if(wl) wl[it] |= 0x10;
                    const int raw = (T0[0] + T0[1] - 1) - S_end;
                    if(out_len + (uint32_t)raw > rw.out_cap) error = LRSC_WALK_ERR_OUTPUT;
                    else {
                        for(int t = 0; t < raw; ++t) out[out_len++] = read[S_end + 1 + t];
                        S_seedLen += raw;
                        correctedLen += T0[1];
                    }
                }
                S_end = T0[0] + T0[1] - 1; S_endBest = T0[5]; S_isRepeat = (T0[3] & 1) != 0; S_maxFixed = T0[2];
                it += 1;
            }
            int next = 0, firstType = 0;
            const int min_SA = a.pb_coverage > 60 ? (int)((a.pb_coverage / 60) * 3) : 3;

            const uint64_t t_all0 = __builtin_readcyclecounter();
            // The lanes of a wavefront stay together in Walk::step(): a lane whose walk has ended handles the result and sets
            // up its next walk (the `!in_walk` branch) while the others wait for it, instead of every lane idling until the
            // longest walk of the wavefront is over.
            uint32_t walks_here = 0;
            const uint64_t steps0 = W.steps;
            bool in_walk = false;
            const int32_t* T = seeds;
            int T_start = 0, T_len = 0, interval = 0, k = 0, trg_len = 0;
            bool T_isRepeat = false, rtou = false;
            while(true) {
              // setting up a walk stalls every stepping lane of the wavefront, so lanes between walks wait until a quorum of
              // the wavefront's live lanes is between walks (lanes that are done leave the loop and stop counting)
              const uint32_t n_all = (uint32_t)__builtin_popcountll(__ballot(true));
              const uint32_t n_need = (uint32_t)__builtin_popcountll(__ballot(!in_walk));
              const bool setup_now = n_need * 100u >= n_all * a.setup_quorum_pct;
              if(!in_walk) {
                if(!(it < n_seeds) || error) break;
                if(next == 0 && a.max_walks != 0 && (walks_here >= a.max_walks || W.steps - steps0 >= (uint64_t)a.max_steps)) { state = kReadYield; break; }
                if(!setup_now) continue;
                ++walks_here;
                const uint64_t t0 = __builtin_readcyclecounter();
                T = seeds + (uint64_t)(it + (uint32_t)next) * kSeedInts;
                T_start = T[0]; T_len = T[1];
                T_isRepeat = (T[3] & 1) != 0;
                interval = T_start - S_end - 1;
                k = (S_endBest < T[4] ? S_endBest : T[4]) - 2;                     // min(source.endBest, target.startBest) - 2
                if(S_isRepeat || T_isRepeat) {
                    k = S_seedLen < T_len ? S_seedLen : T_len;
                    k = k < a.start_kmer_len + 2 ? k : a.start_kmer_len + 2;
                }
                rtou = S_isRepeat && !T_isRepeat;
                trg_len = rtou ? k : T_len;
                if(k < (int)a.seed_size || k > (int)kMaxInitK || k > S_seedLen || interval < 0 || trg_len < (int)a.min_overlap ||
                   (uint32_t)(k + interval + trg_len) > rw.lq_max) { error = LRSC_WALK_ERR_GEOMETRY; break; }
                const uint32_t Lq = (uint32_t)(k + interval + trg_len);
                const uint8_t* tail = out + out_len - k;                           // source.seedStr.substr(seedLen - k)
                if(!rtou) {
                    for(int t = 0; t < k; ++t) q[t] = tail[t];
                    for(int t = 0; t < interval; ++t) q[k + t] = read[S_end + 1 + t];
                    for(int t = 0; t < T_len; ++t) q[k + interval + t] = read[T_start + t];
                } else {
                    // src <-> trg swapped and everything reverse-complemented (:176-184); the walk starts from the
                    // last k characters of revcomp(target seed) = revcomp(target[0..k))
                    for(int t = 0; t < k; ++t) q[t] = (uint8_t)(3 - read[T_start + k - 1 - t]);
                    for(int t = 0; t < interval; ++t) q[k + t] = (uint8_t)(3 - read[S_end + interval - t]);
                    for(int t = 0; t < k; ++t) q[k + interval + t] = (uint8_t)(3 - tail[k - 1 - t]);
                }
                // constructor's bulk part (.cpp:82-94,127-152)
                if(!prepare_all_from_tables<WIDE>(fm, q, Lq, (uint32_t)(k + interval), a.seed_size, a.min_overlap, W.it9f, W.it9r, flags5, term))
                    for(uint32_t i = 0; i < Lq; ++i)
                        prepare_offset<WIDE>(fm, W.sF, W.sR, mtab, q, i, Lq, (uint32_t)(k + interval), a.seed_size, a.min_overlap, W.it9f,
                                             W.it9r, flags5, term, W.n_rank, W.n_blk);
                W.Lq = Lq; W.initk = (uint32_t)k; W.path_len = (uint32_t)interval; W.trg_len = (uint32_t)trg_len; W.dis = interval;
                W.maxOverlap = (uint32_t)k + 2;
                W.min_SA_threshold = (uint64_t)min_SA;
                if(interval > 100) W.maxIndelSize = (uint64_t)(interval * 0.2); else W.maxIndelSize = 20;
                W.maxLength = (uint64_t)((1.2 * (interval + 10)) + (double)(2 * (uint64_t)k));
                W.minLength = (uint64_t)((0.8 * (interval - 20)) + (double)(2 * (uint64_t)k));
                W.n_term = (uint32_t)trg_len - a.min_overlap + 1;
                cyc_prep += __builtin_readcyclecounter() - t0;
                W.begin();
                in_walk = true;
              }
                if(W.step()) continue;
                in_walk = false;
                uint32_t plen = 0, mi = 0;
                const int code = W.finish(&plen, best, &mi);
                if(code <= LRSC_WALK_ERR_CHILDREN) { error = code; break; }
                if(next == 0) firstType = code;

                if(code > 0) {
                    // merged = path + target.substr(i + minOverlap); out = merged (un-reversed) minus its first k characters
                    const uint32_t tail_from = mi + a.min_overlap;
                    const uint32_t tlen = (uint32_t)trg_len - tail_from;
                    const uint32_t M = plen + tlen;
                    uint32_t appended = 0;
                    if(!rtou) {
                        appended = M - (uint32_t)k;
                        if(out_len + appended > rw.out_cap) { error = LRSC_WALK_ERR_OUTPUT; break; }
                        for(uint32_t j = (uint32_t)k; j < M; ++j)
                            out[out_len + j - k] = (uint8_t)(j < plen ? path_get(best, j) : q[k + interval + tail_from + (j - plen)]);
                    } else {
                        // revcomp(merged) + target.substr(k), minus the first k characters (:195-200)
                        const uint32_t total = M + (uint32_t)(T_len - k);
                        appended = total - (uint32_t)k;
                        if(out_len + appended > rw.out_cap) { error = LRSC_WALK_ERR_OUTPUT; break; }
                        for(uint32_t j = (uint32_t)k; j < total; ++j) {
                            uint8_t c;
                            if(j < M) {
                                const uint32_t m = M - 1 - j;                       // index into merged
                                c = (uint8_t)(3 - (m < plen ? path_get(best, m) : q[k + interval + tail_from + (m - plen)]));
                            } else
                                c = read[T_start + k + (j - M)];
                            out[out_len + j - k] = c;
                        }
                    }
                    out_len += appended;
                    correctedLen += appended;
                    seedDis += interval;
                    FMNum++;
                    totalWalkNum++;
                    S_seedLen += (int)appended;                                      // SeedFeature::append
                    S_end = T_start + T_len - 1; S_endBest = T[5]; S_isRepeat = T_isRepeat; S_maxFixed = T[2];
                    it += (uint32_t)next + 1;
                    next = 0;
                    continue;
                }
                if(next + 1 < a.next_target && it + (uint32_t)next + 1 < n_seeds) { next++; continue; }
                switch(firstType) {
                    case -1: highErrorNum++; break;
                    case -2: exceedDepthNum++; break;
                    case -3: exceedLeaveNum++; break;
                    default: error = LRSC_WALK_ERR_CODE; break;
                }
                if(error) break;
                totalWalkNum++;
                if(wl) wl[it] = (uint8_t)((firstType + 4) | (a.no_dp ? 0x10 : 0));
                const int32_t* T0 = seeds + (uint64_t)it * kSeedInts;               // target = *iterTarget
                if(!a.no_dp) {
                    // correctByMSAlignment (:208-236): park the read with its query = src k-mer + raw segment + target seed
                    const int iv0 = T0[0] - S_end - 1;
                    int k0 = (S_endBest < T0[4] ? S_endBest : T0[4]) - 2;
                    if(S_isRepeat || (T0[3] & 1)) {
                        k0 = S_seedLen < T0[1] ? S_seedLen : T0[1];
                        k0 = k0 < a.start_kmer_len + 2 ? k0 : a.start_kmer_len + 2;
                    }
                    if(k0 < 1 || k0 > S_seedLen || k0 > T0[1] || iv0 < 0 || (uint32_t)(k0 + iv0 + T0[1]) > rw.lq_max) { error = LRSC_WALK_ERR_GEOMETRY; break; }
                    const uint8_t* tl = out + out_len - k0;
                    for(int t = 0; t < k0; ++t) dpq[t] = tl[t];
                    for(int t = 0; t < iv0; ++t) dpq[k0 + t] = read[S_end + 1 + t];
                    for(int t = 0; t < T0[1]; ++t) dpq[k0 + iv0 + t] = read[T0[0] + t];
                    R.dp_k = (uint32_t)k0; R.dp_lq = (uint32_t)(k0 + iv0 + T0[1]);
                    R.dp_total_freq = (int64_t)S_maxFixed + (int64_t)T0[2];
                    state = kReadParked;
                    break;
                }
                if(a.split) {
                    if(out_len + (uint32_t)T0[1] > rw.out_cap) { error = LRSC_WALK_ERR_OUTPUT; break; }
                    piece_start[n_pieces++] = out_len;                               // pieceVec.push_back(target)
                    for(int t = 0; t < T0[1]; ++t) out[out_len++] = read[T0[0] + t];
                    S_seedLen = T0[1];
                } else {
                    const int raw = (T0[0] + T0[1] - 1) - S_end;                     // readSeq.substr(source.seedEndPos + 1, target.seedEndPos - source.seedEndPos)
                    if(out_len + (uint32_t)raw > rw.out_cap) { error = LRSC_WALK_ERR_OUTPUT; break; }
                    for(int t = 0; t < raw; ++t) out[out_len++] = read[S_end + 1 + t];
                    S_seedLen += raw;
                }
                correctedLen += T0[1];
                S_end = T0[0] + T0[1] - 1; S_endBest = T0[5]; S_isRepeat = (T0[3] & 1) != 0; S_maxFixed = T0[2];
                it += 1;
                next = 0;
            }
            n_rank += W.n_rank; n_blk += W.n_blk;
            R.steps = W.steps;
            cyc_stitch = __builtin_readcyclecounter() - t_all0 - (cyc_prep - R.cyc[0]) - (W.cyc_setup - R.cyc[1]) - (W.cyc_loop - R.cyc[2]);
            R.cyc[1] = W.cyc_setup; R.cyc[2] = W.cyc_loop;
            R.it = it; R.s_seed_len = S_seedLen; R.s_end = S_end; R.s_end_best = S_endBest; R.s_max_fixed = S_maxFixed;
            R.s_is_repeat = S_isRepeat ? 1 : 0;
        }
        R.c[0] = rlen; R.c[1] = correctedLen; R.c[2] = n_seeds; R.c[3] = totalWalkNum; R.c[4] = highErrorNum;
        R.c[5] = exceedDepthNum; R.c[6] = exceedLeaveNum; R.c[7] = FMNum; R.c[8] = DPNum; R.c[9] = seedDis;
        R.cyc[0] = cyc_prep; R.cyc[3] += cyc_stitch;
        R.n_pieces = n_pieces; R.out_len = out_len; R.merge = n_pieces != 0; R.error = error;
        R.state = error ? kReadDone : state;
    }
    flush_counters(a.ctr, n_rank, n_blk);
}

// Bounds of every walk read r can be asked for: the source always ends where seed it-1 ends and the target is
// seed it + next, next < nextTarget (initCorrect's two loops, :78-157).
__global__ __launch_bounds__(256) void correct_plan_kernel(CorrectArgs a)
{
    const uint32_t r = blockIdx.x * 256 + threadIdx.x;
    if(r >= a.n_reads) return;
    const uint64_t rs = a.read_off[r];
    const uint32_t n_seeds = a.seed_count[r];
    const int32_t* seeds = a.seeds + seed_slab(rs, r, a.min_k) * kSeedInts;
    uint32_t gap_max = 0, lq_max = 0;
    for(uint32_t it = 1; it < n_seeds; ++it) {
        const int s_end = seeds[(uint64_t)(it - 1) * kSeedInts] + seeds[(uint64_t)(it - 1) * kSeedInts + 1] - 1;
        for(int next = 0; next < a.next_target && it + (uint32_t)next < n_seeds; ++next) {
            const int32_t* T = seeds + (uint64_t)(it + (uint32_t)next) * kSeedInts;
            const int gap = T[0] - s_end - 1;
            if(gap < 0) continue;                               // the correction kernel reports it
            if((uint32_t)gap > gap_max) gap_max = (uint32_t)gap;
            const uint32_t lq = kMaxInitK + (uint32_t)gap + (uint32_t)T[1];
            if(lq > lq_max) lq_max = lq;
        }
    }
    a.plan[r].gap_max = gap_max;
    a.plan[r].lq_max = lq_max;
}

__global__ __launch_bounds__(256) void correct_gather_kernel(CorrectArgs a, const uint64_t* dst_off, char* dst)
{
    const uint32_t r = blockIdx.x;
    const uint8_t* src = a.out_codes + a.work[r].out_off;
    char* d = dst + dst_off[r];
    const uint32_t n = (uint32_t)(dst_off[r + 1] - dst_off[r]);            // 0 for a read the host gave up on (per-read status)
    for(uint32_t i = threadIdx.x; i < n; i += 256) d[i] = "ACGT"[src[i] & 3u];
}

hipError_t launch_correct_plan(const CorrectArgs& a, hipStream_t stream)
{
    if(a.n_reads == 0) return hipSuccess;
    hipLaunchKernelGGL(correct_plan_kernel, dim3((a.n_reads + 255) / 256), dim3(256), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_correct_gather(const CorrectArgs& a, const uint64_t* dst_off, char* dst, hipStream_t stream)
{
    if(a.n_reads == 0) return hipSuccess;
    hipLaunchKernelGGL(correct_gather_kernel, dim3(a.n_reads), dim3(256), 0, stream, a, dst_off, dst);
    return hipGetLastError();
}

hipError_t launch_correct_reads(const FmIndexDev& fm, const CorrectArgs& a, hipStream_t stream)
{
    if(a.n_reads == 0) return hipSuccess;
    if(a.reads_per_wave == 0 || a.reads_per_wave > 64 || (a.reads_per_wave & (a.reads_per_wave - 1))) return hipErrorInvalidValue;
    unsigned nb = (a.n_reads + a.reads_per_wave - 1) / a.reads_per_wave;
    if(a.queue && a.queue_waves != 0 && nb > a.queue_waves) nb = a.queue_waves;
    if(a.occupancy >= 4) {
        if(fm.wide) hipLaunchKernelGGL((correct_reads_kernel<true, 4>), dim3(nb), dim3(64), 0, stream, fm, a);
        else        hipLaunchKernelGGL((correct_reads_kernel<false, 4>), dim3(nb), dim3(64), 0, stream, fm, a);
    } else {
        if(fm.wide) hipLaunchKernelGGL((correct_reads_kernel<true, 2>), dim3(nb), dim3(64), 0, stream, fm, a);
        else        hipLaunchKernelGGL((correct_reads_kernel<false, 2>), dim3(nb), dim3(64), 0, stream, fm, a);
    }
    return hipGetLastError();
}

} // namespace lrsc
