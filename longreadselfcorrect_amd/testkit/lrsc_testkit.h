/* lrsc_testkit.h -- NOT part of the drop-in ABI (include/lrsc.h): the synthetic workload generator that bench.py and the
 * tests use, and a test hook into the product's std::sort emulation.  Built into _build/liblrsc_testkit.so. */
#ifndef LRSC_TESTKIT_H
#define LRSC_TESTKIT_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
/* ---- synthetic data (deterministic) -------------------------------- */
/* i.i.d. uniform ACGT genome of `len` bases into out (no terminator). */
int lrsc_synth_genome(uint64_t seed, uint64_t len, char* out);
/* Simulated PacBio reads: template of `tmpl_len` bases at a uniform start, random strand,
 * per-template-base deletion p_del, substitution p_sub, geometric insertion with mean p_ins.
 * out_bases must hold cap bytes; out_off n_reads+1 entries.  Read i depends only on
 * (seed, first_read + i), so shards generate disjoint slices independently.  -6 (capacity) if cap is too small. */
int lrsc_synth_reads(uint64_t seed, const char* genome, uint64_t genome_len,
                     uint64_t first_read, uint32_t n_reads, uint32_t tmpl_len,
                     double p_del, double p_sub, double p_ins,
                     char* out_bases, uint64_t cap, uint64_t* out_off);
/* The permutation std::sort (libstdc++ introsort, comparator a.start > b.start) leaves n (key, index) pairs in -- the
 * product's own re-implementation (csrc/introsort_emul.h), run on the host.  perm_out[j] = index. */
int lrsc_debug_sort_order(const uint64_t* keys, uint32_t n, uint32_t* perm_out);
#ifdef __cplusplus
}
#endif
#endif
