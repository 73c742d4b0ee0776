/**
 *  oracle/sz_oracle.h - plain-C CPU restatement of the reference's batched edit-distance / alignment path.
 *
 *  TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may
 *  compile, link, load or call anything in oracle/.  The product library (stringzilla_amd/csrc) never does,
 *  and fails loudly when its HIP code object is missing instead of falling back to this.
 *
 *  Parity status: PINNED.  tests/test_oracle.py checks every function here against
 *    (1) the reference's own known-answer vectors (test/similarities.cuh:609-625, test/similarities.py:217-226,
 *        284-293) committed under tests/golden/ as JSON,
 *    (2) golden matrices produced by the real reference engines compiled from /root/reference
 *        (oracle/ref_shim.cpp -> oracle/_ref/libszs_ref.so; generator tests/golden/make_golden.py), and
 *    (3) when oracle/_ref is present, live fuzzing against the reference serial engines.
 *
 *  Each function cites the reference file:line whose behaviour it restates.  All arithmetic is integer.
 */
#ifndef SZ_ORACLE_H_
#define SZ_ORACLE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- single-pair scorers -------------------------------------------------------------------------------------- */

/** Dual-row Wagner-Fischer, linear gaps.  Restates test/similarities.cuh:47-70 and the recurrence of
 *  include/stringzillas/similarities/serial.hpp:778-876 (`cell = min(diag + sub, min(top, left) + gap)`). */
uint64_t szo_levenshtein_linear(char const *q, size_t q_len, char const *c, size_t c_len, int8_t match,
                                int8_t mismatch, int8_t gap);

/** Dual-row Gotoh, affine gaps, minimising.  Restates test/similarities.cuh:137-183 and serial.hpp:1002-1137
 *  (finite "discard" seeds `boundary + open + extend`, serial.hpp:1049-1056). */
uint64_t szo_levenshtein_affine(char const *q, size_t q_len, char const *c, size_t c_len, int8_t match,
                                int8_t mismatch, int8_t open, int8_t extend);

/** Bit-parallel Myers/Hyyro, block-based over 64-bit words, unit costs only.  Restates
 *  serial.hpp:2073-2314 (the per-word update at :2182-2204); the pattern is the shorter string. */
uint64_t szo_levenshtein_myers(char const *q, size_t q_len, char const *c, size_t c_len);

/** The dispatcher of serial.hpp:2527-2693: `open == extend` -> linear; unit costs -> Myers. */
uint64_t szo_levenshtein(char const *q, size_t q_len, char const *c, size_t c_len, int8_t match, int8_t mismatch,
                         int8_t open, int8_t extend);

/** `sz_rune_decode_unchecked` over a whole string (include/stringzilla/utf8_runes/serial.h:111-124): returns the rune
 *  count; `runes` needs room for `length` entries.  Bytes missing from a truncated final sequence read as zero. */
size_t szo_utf8_decode(char const *utf8, size_t length, uint32_t *runes);

/** Codepoint-level Levenshtein distance, linear or affine gaps: restates `levenshtein_distance_utf8`,
 *  serial.hpp:2704-2900 (transcode both sides, then the same recurrences over 32-bit symbols). */
uint64_t szo_levenshtein_utf8(char const *q, size_t q_len, char const *c, size_t c_len, int8_t match, int8_t mismatch,
                              int8_t open, int8_t extend);

/** Needleman-Wunsch global score; `open == extend` selects the linear recurrence (needleman_wunsch.cuh:99-118).
 *  Restates test/similarities.cuh:72-98 (linear), :185-229 (Gotoh) and serial.hpp:2910-3007.
 *  cost(a, b) = class_costs[byte_to_class[(u8)a] * 32 + byte_to_class[(u8)b]]   (serial.hpp:199-204). */
int64_t szo_needleman_wunsch(char const *q, size_t q_len, char const *c, size_t c_len, uint8_t const *byte_to_class,
                             int8_t const *class_costs, int8_t open, int8_t extend);

/** Smith-Waterman local score; restates test/similarities.cuh:100-135 (linear), :231-280 (Gotoh) and
 *  serial.hpp:3019-3124 (only the substitution branch is clamped with 0; the best over all cells wins). */
int64_t szo_smith_waterman(char const *q, size_t q_len, char const *c, size_t c_len, uint8_t const *byte_to_class,
                           int8_t const *class_costs, int8_t open, int8_t extend);

/* ---- cross-product drivers (serial.hpp:3140-3184) -------------------------------------------------------------- */
/*  Tapes carry count+1 64-bit offsets.  `c_offsets == NULL` requests symmetric self-similarity: the lower
 *  triangle (incl. diagonal) is scored as (query=i, candidate=j<=i) and mirrored.  results[q*stride + c].        */

void szo_levenshtein_cross(char const *q_data, uint64_t const *q_offsets, size_t q_count, char const *c_data,
                           uint64_t const *c_offsets, size_t c_count, int8_t match, int8_t mismatch, int8_t open,
                           int8_t extend, uint64_t *results, size_t stride);

void szo_levenshtein_utf8_cross(char const *q_data, uint64_t const *q_offsets, size_t q_count, char const *c_data,
                                uint64_t const *c_offsets, size_t c_count, int8_t match, int8_t mismatch, int8_t open,
                                int8_t extend, uint64_t *results, size_t stride);

void szo_needleman_wunsch_cross(char const *q_data, uint64_t const *q_offsets, size_t q_count, char const *c_data,
                                uint64_t const *c_offsets, size_t c_count, uint8_t const *byte_to_class,
                                int8_t const *class_costs, int8_t open, int8_t extend, int64_t *results,
                                size_t stride);

void szo_smith_waterman_cross(char const *q_data, uint64_t const *q_offsets, size_t q_count, char const *c_data,
                              uint64_t const *c_offsets, size_t c_count, uint8_t const *byte_to_class,
                              int8_t const *class_costs, int8_t open, int8_t extend, int64_t *results, size_t stride);

/* ---- cost models ----------------------------------------------------------------------------------------------- */

/** BLOSUM62 folded into the 256-byte map + 32x32 class table form of serial.hpp:193-251 (class 0 = catch-all
 *  costing 0 against everything; only uppercase residues are mapped). */
void szo_blosum62(uint8_t *byte_to_class, int8_t *class_costs);
/** NUC.4.4 in the same compact form (serial.hpp:253-287). */
void szo_nuc44(uint8_t *byte_to_class, int8_t *class_costs);

/** Worst-case reach of serial.hpp:135-162: ((maximise ? q+c : max(q,c)) + (linear ? 1 : 3)) * max(magnitude, 1). */
uint64_t szo_worst_case_reach(size_t q_len, size_t c_len, int maximise, int affine, unsigned magnitude);

/* ---- rolling MinHash / Count-Min fingerprints (sz_oracle_fingerprints.c; reference: fingerprints/serial.hpp) ---- */

/** Per-dimension parameters exactly as the reference seeds them; `window_widths` NULL / 0 = its defaults. */
void szo_fingerprint_parameters(size_t dimensions, size_t const *window_widths, size_t window_widths_count, uint64_t seed,
                                size_t *widths, double *multipliers, double *modulos, double *inverse_modulos,
                                double *negative_discarding_multipliers);

/** `min_hashes` / `min_counts`: `count` rows of `dimensions` consecutive u32. */
void szo_fingerprints_cross(char const *data, uint64_t const *offsets, size_t count, size_t dimensions, size_t alphabet_size,
                            size_t const *window_widths, size_t window_widths_count, uint64_t seed, uint32_t *min_hashes,
                            uint32_t *min_counts);

#ifdef __cplusplus
}
#endif
#endif /* SZ_ORACLE_H_ */
