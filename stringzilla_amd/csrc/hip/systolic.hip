/*
 *  systolic.hip - the FEW-PAIRS tier of the weighted scorers on gfx950: Needleman-Wunsch, Smith-Waterman and weighted /
 *  codepoint-level Levenshtein for cross-products too small to give every lane of the device a pair of its own.
 *
 *  Fills the slot of the reference's intra-pair tiers
 *      (affine_)score_per_cuda_warp_        /root/reference/include/stringzillas/similarities/cuda.cuh:1246-1500
 *      (affine_)score_across_cuda_device_   .../similarities/cuda.cuh:729-1170   (128x128 tiles, `progress[]` counters)
 *  and must return exactly what the serial scorers return (serial.hpp:2527-2693,2910-3124; recurrences :778-1278).
 *
 *  weighted.hip gives every (query, candidate) pair ONE lane.  That is the right shape for a million pairs and the wrong
 *  one for a 1 x 1 call of two 100 KB strings, or for 16 x 16 reads of 4 KB: a handful of lanes would walk the whole
 *  matrix alone (measured: 19 GCUPS on 16 x 16 x 4 KB; this file: 1.9 TCUPS).  Here a pair is spread over wavefronts
 *  instead, as a two-level systolic array:
 *
 *  - A BAND is 64 x R consecutive query rows (R = 8: 512 rows) and belongs to one wavefront; lane l owns rows
 *    [R l, R l + R) of the band, its column of R cells (plus gap tracks) lives in VGPRs, exactly like a strip of
 *    weighted.hip.  A lane scores K = 4 consecutive columns per step (R x K = 32 cells between two exchanges; in the
 *    steady state the four column recurrences are interleaved along the anti-diagonal, so they overlap).
 *  - The lanes of a wavefront are skewed by one step: at step t lane l scores columns K (t - l) ... + K - 1.  What lane l
 *    needs from above - H (and the vertical-gap track) of lane l-1's bottom row under the same columns - was produced by
 *    lane l-1 one step earlier and arrives through `v_mov_b32_dpp wave_shr:1`; the candidate symbols (K class ids or
 *    bytes packed in one dword) travel down the lanes the same way, one step AHEAD of the scores, so that their cost rows
 *    are fetched from LDS a whole step before they are needed.  No barrier, no shuffle through memory.
 *  - Lane 0's inputs (the symbols, the row above the band) are wave-uniform per step: lanes 0..15 preload a chunk of 16
 *    steps (64 columns) one chunk in advance, and `v_readlane_b32` picks the step's values into SGPRs.
 *  - Bands of one pair are chained THROUGH MEMORY: lane 63 parks the band's bottom row [column] and publishes a progress
 *    word every 16 steps; the wavefront of the next band polls that word before it preloads a chunk.  Parked cells and
 *    progress words are agent-scope (`sc1`) accesses - coherent at the device level by themselves - so the hand-over
 *    needs no L2 write-back and no invalidate, only "stores acknowledged before the word is published".
 *  - ONE TICKET PER WAVEFRONT, drawn from an atomic counter in (pair, band) order when the wavefront starts running: the
 *    band a wavefront waits for always holds an EARLIER ticket, i.e. is running or finished - forward progress without
 *    any co-residency requirement.  All bands of a long pair are thus in flight at once, each trailing its predecessor
 *    by ~95 steps (63 of lane skew + 2 chunks of hand-over).
 *  - The control words (ticket counter, progress, local-alignment best / done) are tagged with the launch's EPOCH in
 *    their high half: a word of an older launch compares below everything of this one, so nothing is ever cleared
 *    between launches and no launch depends on a fill having landed.  A wait that cannot be satisfied (a broken
 *    invariant, never observed) gives up after a fraction of a second and flags the call instead of hanging the device.
 *  - Substitution costs: class-table engines build a per-band profile in LDS, profile[class][lane] = the R int8 costs
 *    of the lane's rows against that class (conflict-free ds_read_b64; the candidate is mapped to classes when the chunk
 *    is preloaded).  Uniform-cost engines (weighted / codepoint Levenshtein) compare the symbol with the lane's R query
 *    symbols held in registers - bytes and UTF-32 runes alike.
 *
 *  What bounds it (profiles/r01/wave_latency.json, shapes_v6.jsonl): a wavefront that has its SIMD to itself issues a
 *  dependent instruction every ~8 cycles, a step is ~170 instructions = ~1400 cycles (2700 with affine gaps), and a pair
 *  needs len(candidate) / K + 63 + 95 (bands - 1) steps end to end.  R = 8, K = 4 sits at the minimum of that product.
 *
 *  Borders, the finite affine "discard" seeds and the clamp of local alignment follow weighted.hip (and through it
 *  serial.hpp:821-823,1045-1056,1195-1201,957-965) to the letter; tests pin both tiers against the same oracle.
 */
#include "device_common.hpp"

namespace szs_hip {

#ifndef SZS_SYSTOLIC_ROWS
#define SZS_SYSTOLIC_ROWS 8
#endif
constexpr int systolic_rows_k = SZS_SYSTOLIC_ROWS;              // R: query rows per lane
constexpr u32 systolic_band_rows_k = 64u * systolic_rows_k;     // query rows per band = per wavefront
#ifndef SZS_SYSTOLIC_WAVES
#define SZS_SYSTOLIC_WAVES 4
#endif
constexpr u32 systolic_waves_k = SZS_SYSTOLIC_WAVES;            // wavefronts per workgroup; each pulls its own tickets
/** A parked cell is written by one wavefront and read by another one, usually on another XCD.  Both sides use
 *  agent-scope accesses (`sc1`: the cell lives at the device's coherence point, not in an XCD's write-back L2), so the
 *  release / acquire around a chunk has nothing left to write back or to re-fetch - measured 1.5x faster on a 16 x 16
 *  batch of 4 KB strings than plain stores flushed by `buffer_wbl2`, and one assumption fewer about the caches. */
__device__ __forceinline__ i32 parked_load(i32 const *cell) {
    return __hip_atomic_load(cell, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void parked_store(i32 *cell, i32 value) {
    __hip_atomic_store(cell, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
constexpr u32 systolic_columns_k = 4;                           // K: consecutive columns a lane scores per step
constexpr u32 systolic_chunk_steps_k = 16;                      // steps (of K columns) per hand-over between bands
constexpr u32 systolic_slack_k = 64;                            // parked columns past the longest candidate
constexpr size_t systolic_header_bytes_k = 256;                 // ticket counter [0] and stall flag [1] at the head of the control block
constexpr unsigned long long systolic_patience_ticks_k = 200000000ull; // 2 s of the 100 MHz wall clock: how long a band waits for its
                                                                     // predecessor before it flags the call - ELAPSED time, not a poll count: a slow predecessor
                                                                     // (a shared GPU, a profiler) is not a stall; the host re-runs a flagged call on the lanes tier
static_assert(systolic_rows_k % 4 == 0 && systolic_rows_k <= 16, "R int8 costs are fetched as one LDS read");
static_assert(systolic_band_rows_k == SZS_SYSTOLIC_BAND_ROWS, "the host planner models bands of this height");

__device__ __forceinline__ i32 smax2(i32 a, i32 b) { return a > b ? a : b; }
__device__ __forceinline__ i32 smax3(i32 a, i32 b, i32 c) { return smax2(smax2(a, b), c); }

/** `value` of lane l-1 for every lane l >= 1; lane 0 receives `first` (wave-uniform).  One v_mov_b32_dpp wave_shr:1. */
__device__ __forceinline__ u32 from_lane_above(u32 first, u32 value) {
    return (u32)__builtin_amdgcn_update_dpp((int)first, (int)value, 0x138 /* wave_shr:1 */, 0xF, 0xF, false);
}
__device__ __forceinline__ i32 from_lane_above(i32 first, i32 value) {
    return __builtin_amdgcn_update_dpp(first, value, 0x138, 0xF, 0xF, false);
}

/** See weighted.hip `gapped`: plain signed add, or - local alignment with non-positive gap costs, where every value a
 *  gap applies to is >= 0 - one unsigned saturating subtract of the penalty. */
template <bool saturating_>
__device__ __forceinline__ i32 systolic_gapped(i32 value, i32 gap) {
    if constexpr (saturating_) return (i32)__builtin_elementwise_sub_sat((u32)value, (u32)(-gap));
    else return value + gap;
}

/** The lane's column of R cells in the form the recurrences consume them (weighted.hip `strip_column_t`). */
template <bool affine_>
struct systolic_column_t {
    i32 h[systolic_rows_k];
    i32 h_gapped[systolic_rows_k];
    i32 across_extended[affine_ ? systolic_rows_k : 1];
};

/**
 *  One lane, one column: the recurrences of serial.hpp:846-848 (linear) and :1091-1102 (Gotoh), local alignment clamping
 *  only the substitution branch (:957-965, :1238-1239).  `cost_of(r)` yields the substitution cost of row r.
 */
template <bool local_, bool affine_, bool saturating_, typename cost_of_t>
__device__ __forceinline__ void systolic_advance(systolic_column_t<affine_> &column, cost_of_t cost_of, i32 above_h,
                                                 i32 above_down, i32 &diagonal, i32 gap_open, i32 gap_extend,
                                                 i32 &down_out, i32 &best, u32 counted_rows) {
    constexpr int rows = systolic_rows_k;
    i32 diag = diagonal;
    diagonal = above_h;
    i32 above_gapped = systolic_gapped<saturating_>(above_h, gap_open);
    i32 down_extended = affine_ ? systolic_gapped<saturating_>(above_down, gap_extend) : 0;
    i32 down = 0;
#pragma unroll
    for (int r = 0; r < rows; ++r) {
        i32 substituted = diag + cost_of(r);
        if constexpr (local_ && !saturating_) substituted = smax2(substituted, 0);
        diag = column.h[r];
        i32 cell;
        if constexpr (affine_) {
            i32 const across = smax2(column.h_gapped[r], column.across_extended[r]);
            down = smax2(above_gapped, down_extended);
            cell = smax3(down, across, substituted);
            column.across_extended[r] = systolic_gapped<saturating_>(across, gap_extend);
            down_extended = systolic_gapped<saturating_>(down, gap_extend);
        }
        else { cell = smax3(above_gapped, column.h_gapped[r], substituted); }
        column.h[r] = cell;
        above_gapped = systolic_gapped<saturating_>(cell, gap_open);
        column.h_gapped[r] = above_gapped;
    }
    down_out = down;
    if constexpr (local_) {
        if (counted_rows >= (u32)rows) {
#pragma unroll
            for (int r = 0; r < rows; r += 2) best = smax3(best, column.h[r], column.h[r + 1]);
        }
        else { // lanes of the last band that hold padded rows: those never count
#pragma unroll
            for (int r = 0; r < rows; ++r)
                if ((u32)r < counted_rows) best = smax2(best, column.h[r]);
        }
    }
}

/**
 *  One lane, K = 4 consecutive columns, interleaved: iteration i scores row i of the first column, row i - 1 of the second,
 *  ... - four cells on one anti-diagonal, none depending on another - so a wavefront alone on its SIMD overlaps the latency
 *  of one column's chain with the work of the other three (scripts/wave_latency.hip: 8.1 cycles per dependent instruction
 *  against 4.6 with four chains in flight).  Same recurrences, same in-place column state as `systolic_advance` called
 *  K times: column j + 1 reads row r only after column j has written it, and keeps its own diagonal like the sequential form.
 *  Every row counts for `best` (the caller takes the sequential form for the one band with padded rows).
 *  `cost_of(j, r)`: substitution cost of row r against the symbol of column j.
 */
template <bool local_, bool affine_, bool saturating_, typename cost_of_t>
__device__ __forceinline__ void systolic_advance_interleaved(systolic_column_t<affine_> &column, cost_of_t cost_of,
                                                             i32 const (&above_h)[systolic_columns_k],
                                                             i32 const (&above_down)[systolic_columns_k], i32 &diagonal,
                                                             i32 gap_open, i32 gap_extend,
                                                             i32 (&bottom_h)[systolic_columns_k],
                                                             i32 (&bottom_down)[affine_ ? systolic_columns_k : 1], i32 &best) {
    constexpr int rows = systolic_rows_k, columns = (int)systolic_columns_k;
    i32 diag[columns], above_gapped[columns], down_extended[columns], down[columns];
#pragma unroll
    for (int j = 0; j < columns; ++j) {
        diag[j] = j == 0 ? diagonal : above_h[j ? j - 1 : 0];
        above_gapped[j] = systolic_gapped<saturating_>(above_h[j], gap_open);
        down_extended[j] = affine_ ? systolic_gapped<saturating_>(above_down[j], gap_extend) : 0;
        down[j] = 0;
    }
    diagonal = above_h[columns - 1];
#pragma unroll
    for (int i = 0; i < rows + columns - 1; ++i) {
        i32 fresh[columns];
#pragma unroll
        for (int j = 0; j < columns; ++j) {
            fresh[j] = 0; // local scores are >= 0: a neutral element for the running maximum
            int const r = i - j;
            if (r < 0 || r >= rows) continue;
            i32 substituted = diag[j] + cost_of(j, r);
            if constexpr (local_ && !saturating_) substituted = smax2(substituted, 0);
            diag[j] = column.h[r];
            i32 cell;
            if constexpr (affine_) {
                i32 const across = smax2(column.h_gapped[r], column.across_extended[r]);
                down[j] = smax2(above_gapped[j], down_extended[j]);
                cell = smax3(down[j], across, substituted);
                column.across_extended[r] = systolic_gapped<saturating_>(across, gap_extend);
                down_extended[j] = systolic_gapped<saturating_>(down[j], gap_extend);
            }
            else { cell = smax3(above_gapped[j], column.h_gapped[r], substituted); }
            column.h[r] = cell;
            above_gapped[j] = systolic_gapped<saturating_>(cell, gap_open);
            column.h_gapped[r] = above_gapped[j];
            fresh[j] = cell;
            if (r == rows - 1) {
                bottom_h[j] = cell;
                if constexpr (affine_) bottom_down[affine_ ? j : 0] = down[j];
            }
        }
        if constexpr (local_) {
            best = smax3(best, fresh[0], fresh[1]);
            best = smax3(best, fresh[2], fresh[3]);
        }
    }
}

/**
 *  @tparam local_       Smith-Waterman instead of a global alignment.
 *  @tparam affine_      Gotoh's three tracks instead of one.
 *  @tparam uniform_     (match, mismatch) costs on raw symbols - Levenshtein engines, maximising negated costs - instead
 *                       of the 32x32 class table.
 *  @tparam runes_       (with uniform_) strings are UTF-32 arrays, lengths count runes.
 *  @tparam saturating_  (with local_) both gap costs <= 0: unsigned-saturating gap arithmetic.
 *
 *  Control block (epoch-tagged 64-bit words, see hip/kernels.h) and parked rows:
 *    work_counter[0], [1]               ticket counter, stall flag
 *    progress[pair * max_bands + band]  columns of that band's bottom row that are parked and visible
 *    pair_best[pair], pair_done[pair]   local alignment: running maximum and finished bands of the pair
 *    parked[pair][plane][column]        bottom rows in flight (plane 0: H, plane 1: vertical-gap track), reused IN PLACE
 *                                       by successive bands: a band overwrites a column 63 steps after it consumed it.
 */
template <bool local_, bool affine_, bool uniform_, bool runes_, bool saturating_>
__global__ __launch_bounds__(64 * systolic_waves_k) void systolic_scores_kernel(
    szs_cost_model_t const *__restrict__ model, szs_string_ref_t const *__restrict__ queries, u32 queries_count,
    szs_string_ref_t const *__restrict__ candidates, u32 candidates_count, u32 max_bands, i64 *__restrict__ results,
    u64 results_row_stride, int symmetric, u64 *__restrict__ work_counter, u64 *progress, u64 *pair_best, u64 *pair_done,
    i32 *parked, u32 parked_columns, u32 epoch) {

    constexpr int rows = systolic_rows_k;
    constexpr int cost_dwords = rows / 4;
    constexpr u32 K = systolic_columns_k;
    constexpr u32 chunk_steps = systolic_chunk_steps_k;
    static_assert(!runes_ || uniform_, "codepoint scoring exists for uniform costs only");
    static_assert(!saturating_ || local_, "saturating gap arithmetic is a local-alignment form");
    static_assert(K == 4, "byte / class symbols of one step travel packed in one dword");

    // Class-table engines: [wave][class][lane] packed int8 costs of the lane's rows; shared: the table and the byte map.
    __shared__ __attribute__((aligned(16))) u32 profiles[uniform_ ? 1 : systolic_waves_k * 32 * 64 * cost_dwords];
    __shared__ int8_t table[uniform_ ? 1 : 32 * 32];
    __shared__ u8 class_of_byte[uniform_ ? 1 : 256];

    u32 const lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    i32 const gap_open = model->gap_open, gap_extend = model->gap_extend;
    i32 const uniform_match = model->uniform_match, uniform_mismatch = model->uniform_mismatch;
    if constexpr (!uniform_) {
        class_of_byte[threadIdx.x] = model->byte_to_class[threadIdx.x];
        for (u32 i = threadIdx.x; i < 32 * 32; i += 64 * systolic_waves_k) table[i] = (int8_t)model->substitution[i];
        __syncthreads(); // the only barrier: from here on every wavefront runs on its own
    }
    u32 *const profile = profiles + (uniform_ ? 0 : wave * 32 * 64 * cost_dwords) + (uniform_ ? 0 : lane * cost_dwords);

    u64 const total_tickets = (u64)queries_count * candidates_count * max_bands;
    u64 const tag = (u64)epoch << 32;
    u32 const planes = affine_ ? 2 : 1;

    // ---- one ticket per wavefront: the grid holds exactly as many wavefronts as there are tickets.
    // Every word of the control block is tagged with the launch's epoch in its high half, so NOTHING in it has to be zeroed
    // between launches (and no launch depends on a preceding fill having landed): the first fetch-max lifts a word left by
    // an older launch to (epoch, 0), values of older epochs compare below everything of this one.
    u32 ticket = 0;
    if (lane == 0) {
        __hip_atomic_fetch_max(work_counter, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ticket = (u32)__hip_atomic_fetch_add(work_counter, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    ticket = __builtin_amdgcn_readfirstlane(ticket);
    if (ticket >= total_tickets) return;
    u32 const pair = ticket / max_bands, band = ticket % max_bands;
    szs_string_ref_t const query = queries[pair / candidates_count];
    szs_string_ref_t const candidate = candidates[pair % candidates_count];
    if ((symmetric & SZS_LAYOUT_SYMMETRIC) && candidate.index > query.index) return; // upper triangle: mirrored from below
    u32 const m = query.length, n = candidate.length;
    u32 const bands = m ? (m + systolic_band_rows_k - 1) / systolic_band_rows_k : 1;
    if (band >= bands) return;

    // All-gap borders (weighted.hip; serial.hpp:821-823,1045-1047): DP cell (k, 0) and (0, k).
    auto border = [&](u32 k) -> i32 {
        if constexpr (local_) return 0;
        if constexpr (affine_) return k ? gap_open + gap_extend * (i32)(k - 1) : 0;
        return gap_open * (i32)k;
    };
    auto write_result = [&](i32 score) {
        i64 const value = uniform_ ? -(i64)score : (i64)score;
        bool const transposed = (symmetric & SZS_LAYOUT_TRANSPOSED) != 0; // kernel roles swapped by the host
        u64 const row = transposed ? candidate.index : query.index, column = transposed ? query.index : candidate.index;
        results[row * results_row_stride + column] = value;
        if ((symmetric & SZS_LAYOUT_SYMMETRIC) && candidate.index != query.index)
            results[column * results_row_stride + row] = value;
    };
    if (m == 0 || n == 0) { // an empty side never enters the column loop: the score is the border itself
        if (lane == 0) write_result(local_ ? 0 : border(m ? m : n));
        return;
    }

    bool const first_band = band == 0, last_band = band + 1 == bands;
    u32 const band_first = band * systolic_band_rows_k;
    u32 const first_row = band_first + lane * rows;                                // 0-based string row
    u32 const my_rows = first_row >= m ? 0u : (m - first_row < (u32)rows ? m - first_row : (u32)rows);

    // ---- the lane's query rows: classes folded into the LDS profile, or raw symbols kept in registers
    u32 query_symbols[uniform_ ? rows : 1];
    if constexpr (uniform_) {
#pragma unroll
        for (int r = 0; r < rows; ++r) {
            u32 symbol = ~0u; // padded rows: equal to no byte and to no decoded rune
            if ((u32)r < my_rows)
                symbol = runes_ ? reinterpret_cast<u32 const *>(query.address)[first_row + r]
                                : (u32) reinterpret_cast<u8 const *>(query.address)[first_row + r];
            query_symbols[r] = symbol;
        }
    }
    else {
        u32 classes[rows];
#pragma unroll
        for (int r = 0; r < rows; ++r)
            classes[r] = (u32)r < my_rows ? (u32)class_of_byte[reinterpret_cast<u8 const *>(query.address)[first_row + r]] : 0u;
        for (u32 candidate_class = 0; candidate_class < 32; ++candidate_class) {
            u32 packed[cost_dwords];
#pragma unroll
            for (int r = 0; r < rows; ++r) {
                // cost(query, candidate) = table[class(query)][class(candidate)]: the QUERY picks the row
                i32 const cost = (u32)r < my_rows ? (i32)table[classes[r] * 32 + candidate_class] : 0;
                if (r % 4 == 0) packed[r / 4] = 0;
                packed[r / 4] |= ((u32)cost & 0xFFu) << (8 * (r % 4));
            }
#pragma unroll
            for (int d = 0; d < cost_dwords; ++d) profile[candidate_class * 64 * cost_dwords + d] = packed[d];
        }
    }

    // ---- column 0 of the lane's rows; finite "discard" seeds of the gap tracks (serial.hpp:1049-1056,1195-1201)
    systolic_column_t<affine_> column;
#pragma unroll
    for (int r = 0; r < rows; ++r) {
        column.h[r] = border(first_row + r + 1);
        column.h_gapped[r] = systolic_gapped<saturating_>(column.h[r], gap_open);
        if constexpr (affine_)
            column.across_extended[r] = saturating_ ? 0 : column.h[r] + gap_open + gap_extend + gap_extend;
    }
    i32 diagonal = border(first_row); // DP cell (row above the lane's first, column - 1)
    i32 best = 0, down_out = 0;
    // this lane's bottom row at the K columns of its latest step: what the lane below consumes one step later
    i32 bottom_h[K] = {0, 0, 0, 0}, bottom_down[affine_ ? K : 1] = {0};

    // [1-based DP column]: the row this band parks IS the row its predecessor parked - a band overwrites a column 63
    // steps after it consumed it, and nobody but its successor reads it afterwards.
    i32 *const parked_h = parked + (u64)pair * planes * parked_columns;
    i32 *const parked_down = parked_h + parked_columns;
    u64 *const progress_out = progress + (u64)pair * max_bands + band;
    u64 const *const progress_in = progress_out - 1;

    // The symbols of one step: K class ids / bytes packed in one dword, or K runes.
    constexpr int symbol_words = runes_ ? (int)K : 1;
    struct step_symbols_t {
        u32 word[symbol_words];
    };
    auto load_step_symbols = [&](u32 step_index) -> step_symbols_t { // columns K step_index .. K step_index + K - 1; 0 past the text
        step_symbols_t symbols;
#pragma unroll
        for (int w = 0; w < symbol_words; ++w) symbols.word[w] = 0;
#pragma unroll
        for (u32 j = 0; j < K; ++j) {
            u32 const index = K * step_index + j;
            if (index < n) {
                if constexpr (runes_) symbols.word[j] = reinterpret_cast<u32 const *>(candidate.address)[index];
                else {
                    u32 symbol = reinterpret_cast<u8 const *>(candidate.address)[index];
                    if constexpr (!uniform_) symbol = class_of_byte[symbol];
                    symbols.word[0] |= symbol << (8 * j);
                }
            }
        }
        return symbols;
    };
    auto symbol_of = [&](step_symbols_t const &symbols, u32 j) -> u32 {
        if constexpr (runes_) return symbols.word[j];
        else return (symbols.word[0] >> (8 * j)) & 0xFFu;
    };

    // Lane 0's inputs are preloaded a chunk (16 steps = 64 columns; lane k < 16 holds step k's K columns) at a time and
    // picked per step by readlane: `chunk_*` feed the steps of the current chunk, `next_*` were loaded one chunk EARLIER
    // for the chunk after it, so neither the text nor the parked row is ever waited for inside a chunk.  The price is
    // that a band only starts a chunk when its predecessor has parked the NEXT chunk as well.
    step_symbols_t chunk_symbols = load_step_symbols(0), next_symbols = load_step_symbols(lane < chunk_steps ? lane : 0);
    i32 chunk_above[K] = {0, 0, 0, 0}, next_above[K] = {0, 0, 0, 0};
    i32 chunk_down[affine_ ? K : 1] = {0}, next_down[affine_ ? K : 1] = {0};
    u64 parked_seen = 0;    // the predecessor's progress word (epoch, columns) as last read
    bool abandoned = false; // a wait of this band has timed out: its results are void, never wait again
    auto preload_above = [&](u32 first_step) { // the predecessor's bottom row under the steps [first_step, first_step + 16)
        u32 const last_column = K * (first_step + chunk_steps);
        u64 const needed = tag | (last_column < n ? last_column : n);
        unsigned long long wait_started = 0;
        for (u32 spins = 0; parked_seen < needed && !abandoned; ++spins) {
            parked_seen = __hip_atomic_load(progress_in, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (parked_seen >= needed) break;
            __builtin_amdgcn_s_sleep(2);
            // A predecessor only ever needs to score a few hundred columns to satisfy this wait (microseconds).  Never
            // hang the device on a broken invariant: give up after two seconds - at once if another band
            // already has - flag the call, and let the host report the failure.
            if (spins == 0) wait_started = wall_clock64();
            bool const hopeless = (spins % 256 == 255 && wall_clock64() - wait_started > systolic_patience_ticks_k) ||
                                  (spins % 1024 == 1023 && __hip_atomic_load(work_counter + 1, __ATOMIC_RELAXED,
                                                                             __HIP_MEMORY_SCOPE_AGENT) == (tag | 1));
            if (hopeless) {
                if (lane == 0) __hip_atomic_fetch_max(work_counter + 1, tag | 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                abandoned = true;
            }
        }
        // Parked cells and progress words are all agent-scope (sc1) accesses, coherent by themselves: nothing has to be
        // invalidated here (an agent-scope acquire would `buffer_inv sc1` the XCD's L2 under everybody's feet every 16
        // steps).  The loads below are issued after the counter has been SEEN, which is all the ordering they need.
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        if (lane < chunk_steps) {
#pragma unroll
            for (u32 j = 0; j < K; ++j) {
                u32 const index = K * (first_step + lane) + j;
                if (index < n) {
                    next_above[j] = parked_load(parked_h + index + 1);
                    if constexpr (affine_) next_down[j] = parked_load(parked_down + index + 1);
                }
            }
        }
    };
    if (!first_band) preload_above(0);
    u32 const column_steps = (n + K - 1) / K; // steps a lane needs for the whole text
    bool const every_row_counts = !last_band || m - band_first >= systolic_band_rows_k; // no lane of this band holds padded rows
    u32 const steps = column_steps + 63;      // lane l is busy during steps [l, l + column_steps)

    // The symbol pipeline runs ONE STEP AHEAD of the score pipeline: `symbols_ahead` of lane l holds the symbols of the
    // columns the lane scores next step, so their profile rows are fetched from LDS a whole step before they are consumed.
    step_symbols_t symbols_ahead;
#pragma unroll
    for (int w = 0; w < symbol_words; ++w) symbols_ahead.word[w] = 0;
    u32 costs_ahead[uniform_ ? 1 : K][uniform_ ? 1 : cost_dwords];
    auto advance_symbols = [&](step_symbols_t const &fed) {
#pragma unroll
        for (int w = 0; w < symbol_words; ++w) symbols_ahead.word[w] = from_lane_above(fed.word[w], symbols_ahead.word[w]);
        if constexpr (!uniform_) { // always valid classes: everything ever fed is a class id or 0
#pragma unroll
            for (u32 j = 0; j < K; ++j) {
                u32 const *const row = profile + symbol_of(symbols_ahead, j) * (64 * cost_dwords);
                if constexpr (cost_dwords == 2) {
                    uint2 const both = *reinterpret_cast<uint2 const *>(row);
                    costs_ahead[j][0] = both.x, costs_ahead[j][1] = both.y;
                }
                else if constexpr (cost_dwords == 4) {
                    uint4 const all = *reinterpret_cast<uint4 const *>(row);
                    costs_ahead[j][0] = all.x, costs_ahead[j][1] = all.y, costs_ahead[j][2] = all.z, costs_ahead[j][3] = all.w;
                }
                else { costs_ahead[j][0] = row[0]; }
            }
        }
    };
    auto pick = [&](step_symbols_t const &from, u32 slot) -> step_symbols_t { // lane `slot`'s copy, wave-uniform
        step_symbols_t picked;
#pragma unroll
        for (int w = 0; w < symbol_words; ++w) picked.word[w] = (u32)__builtin_amdgcn_readlane((int)from.word[w], (int)slot);
        return picked;
    };

    // One step of the whole wavefront: every busy lane scores K columns.  `slot` = step % 16 selects lane 0's inputs.
    // `predicated`: some lane is idle or past the end of the text during this chunk; `topmost`: the band's row above is
    // the DP border, not a parked row.  Both are compile-time variants so that the common body - every lane busy, K whole
    // columns - is straight-line code: four independent column recurrences the scheduler can interleave, which is worth
    // a factor of 1.7 to a wavefront that has its SIMD to itself (scripts/wave_latency.hip: 8.1 cycles per dependent
    // instruction, 4.6 with independent work in between).
    auto step = [&](u32 t, u32 slot, auto predicated, auto topmost) {
        constexpr bool is_predicated = decltype(predicated)::value, is_topmost = decltype(topmost)::value;
        step_symbols_t const symbols = symbols_ahead; // of columns K (t - lane) ...
        u32 packed[uniform_ ? 1 : K][uniform_ ? 1 : cost_dwords];
        if constexpr (!uniform_) {
#pragma unroll
            for (u32 j = 0; j < K; ++j)
#pragma unroll
                for (int d = 0; d < cost_dwords; ++d) packed[j][d] = costs_ahead[j][d];
        }
        // feed step t + 1 into the symbol pipeline: the last slot of a chunk takes it from the next chunk's preload
        advance_symbols(slot + 1 == chunk_steps ? pick(next_symbols, 0) : pick(chunk_symbols, slot + 1));

        i32 above_h[K], above_down[K];
#pragma unroll
        for (u32 j = 0; j < K; ++j) {
            i32 fed_above, fed_down = 0;
            if constexpr (is_topmost) {
                fed_above = border(K * t + j + 1);
                if constexpr (affine_) fed_down = saturating_ ? 0 : fed_above + gap_open + gap_extend;
            }
            else {
                fed_above = __builtin_amdgcn_readlane(chunk_above[j], (int)slot);
                if constexpr (affine_) fed_down = __builtin_amdgcn_readlane(chunk_down[affine_ ? j : 0], (int)slot);
            }
            above_h[j] = from_lane_above(fed_above, bottom_h[j]);
            above_down[j] = 0;
            if constexpr (affine_) above_down[j] = from_lane_above(fed_down, bottom_down[affine_ ? j : 0]);
        }
        u32 const my_step = t - lane; // wraps for lanes that have not started yet
        bool const busy = is_predicated ? my_step < column_steps : true;
        if (busy) {
            bool interleaved = false;
            if constexpr (!is_predicated) {
                if (!local_ || every_row_counts) { // K whole columns: the interleaved form
                    interleaved = true;
                    if constexpr (uniform_) {
                        u32 step_symbol[K];
#pragma unroll
                        for (u32 j = 0; j < K; ++j) step_symbol[j] = symbol_of(symbols, j);
                        auto cost_of = [&](int j, int r) -> i32 {
                            return query_symbols[r] == step_symbol[j] ? uniform_match : uniform_mismatch;
                        };
                        systolic_advance_interleaved<local_, affine_, saturating_>(column, cost_of, above_h, above_down, diagonal,
                                                                                   gap_open, gap_extend, bottom_h, bottom_down, best);
                    }
                    else {
                        auto cost_of = [&](int j, int r) -> i32 { return (i32)(int8_t)(packed[j][r / 4] >> (8 * (r % 4))); };
                        systolic_advance_interleaved<local_, affine_, saturating_>(column, cost_of, above_h, above_down, diagonal,
                                                                                   gap_open, gap_extend, bottom_h, bottom_down, best);
                    }
                }
            }
            if (!interleaved) {
#pragma unroll
                for (u32 j = 0; j < K; ++j) {
                    bool const inside = is_predicated ? K * my_step + j < n : true; // the last step may be ragged
                    if (inside) {
                        if constexpr (uniform_) {
                            u32 const symbol = symbol_of(symbols, j);
                            auto cost_of = [&](int r) -> i32 { return query_symbols[r] == symbol ? uniform_match : uniform_mismatch; };
                            systolic_advance<local_, affine_, saturating_>(column, cost_of, above_h[j], above_down[j], diagonal,
                                                                           gap_open, gap_extend, down_out, best, my_rows);
                        }
                        else {
                            auto cost_of = [&](int r) -> i32 { return (i32)(int8_t)(packed[j][r / 4] >> (8 * (r % 4))); };
                            systolic_advance<local_, affine_, saturating_>(column, cost_of, above_h[j], above_down[j], diagonal,
                                                                           gap_open, gap_extend, down_out, best, my_rows);
                        }
                        bottom_h[j] = column.h[rows - 1];
                        if constexpr (affine_) bottom_down[j] = down_out;
                    }
                }
            }
            if (!last_band && lane == 63) { // the band's bottom row under this step's columns, 1-based DP column
#pragma unroll
                for (u32 j = 0; j < K; ++j) {
                    if (is_predicated && K * my_step + j >= n) continue;
                    parked_store(parked_h + K * my_step + j + 1, bottom_h[j]);
                    if constexpr (affine_) parked_store(parked_down + K * my_step + j + 1, bottom_down[affine_ ? j : 0]);
                }
            }
        }
        // Publish the parked columns every 16 steps of lane 63 and at the end of the text.
        if (!last_band && t >= 63) {
            u32 const parked_steps = t - 62;
            if ((parked_steps % chunk_steps == 0 || t + 1 == steps) && lane == 63) {
                u32 const parked_count = K * parked_steps < n ? K * parked_steps : n;
                // wait for the parked cells (sc1 stores: acknowledged at the device's coherence point) - no L2 write-back
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); // compiler ordering
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // every parked store of this lane has been acknowledged
                __hip_atomic_store(progress_out, tag | parked_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    };
    auto run_chunk = [&](u32 chunk_first, auto topmost) {
        // every lane busy, with K whole columns, during all 16 steps?
        bool const steady = chunk_first >= 64 && K * (chunk_first + chunk_steps) <= n;
        if (steady) {
#pragma unroll 1
            for (u32 slot = 0; slot < chunk_steps; ++slot) step(chunk_first + slot, slot, std::false_type {}, topmost);
        }
        else {
            u32 const stop = steps - chunk_first < chunk_steps ? steps - chunk_first : chunk_steps;
#pragma unroll 1
            for (u32 slot = 0; slot < stop; ++slot) step(chunk_first + slot, slot, std::true_type {}, topmost);
        }
    };

    for (u32 chunk_first = 0; chunk_first < steps; chunk_first += chunk_steps) {
        // ---- lane 0 is about to consume the steps [chunk_first, chunk_first + 16)
        if (K * chunk_first < n) {
            chunk_symbols = next_symbols;
#pragma unroll
            for (u32 j = 0; j < K; ++j) {
                chunk_above[j] = next_above[j];
                if constexpr (affine_) chunk_down[j] = next_down[j];
            }
            next_symbols = load_step_symbols(chunk_first + chunk_steps + (lane < chunk_steps ? lane : 0)); // retires under the steps below
            if (!first_band && K * (chunk_first + chunk_steps) < n) preload_above(chunk_first + chunk_steps);
            if (chunk_first == 0) advance_symbols(pick(chunk_symbols, 0)); // the columns of step 0
        }
        if (first_band) run_chunk(chunk_first, std::true_type {});
        else run_chunk(chunk_first, std::false_type {});
    }

    // ---- the pair's score
    if constexpr (local_) {
        u32 const wave_best = wave_max_u32((u32)best); // local scores are >= 0
        if (bands == 1) {
            if (lane == 0) write_result((i32)wave_best);
        }
        else if (lane == 0) { // the band that finishes last reports the maximum over all of them
            __hip_atomic_fetch_max(pair_best + pair, tag | wave_best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_max(pair_done + pair, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            u32 const finished =
                (u32)__hip_atomic_fetch_add(pair_done + pair, 1ull, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            if (finished + 1 == bands)
                write_result((i32)(u32)__hip_atomic_load(pair_best + pair, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        }
    }
    else if (last_band) { // bottom-right cell: the last real row, frozen at the owning lane's last column
        u32 const last_row = m - 1 - band_first;
        i32 mine = 0;
#pragma unroll
        for (int r = 0; r < rows; ++r)
            if ((u32)r == last_row % rows) mine = column.h[r];
        i32 const score = __builtin_amdgcn_readlane(mine, (int)(last_row / rows));
        if (lane == 0) write_result(score);
    }
}

/** One wavefront per ticket.  Wavefronts the device cannot hold yet simply start later; a wavefront draws its ticket
 *  when it starts RUNNING, so the holder of every earlier ticket is running or done - the forward-progress argument of
 *  the band chain does not care how many workgroups are resident. */
static u32 systolic_grid(u64 tickets) { return (u32)((tickets + systolic_waves_k - 1) / systolic_waves_k); }

/** Two blocks: `control` (epoch-tagged 64-bit words: ticket counter, stall flag, progress, best, done - zeroed ONCE when
 *  the block is allocated, never between launches) and `parked` (the bottom rows in flight; plain data). */
struct systolic_layout_t {
    u64 pairs, tickets;
    u32 max_bands, parked_columns;
    size_t progress_at, best_at, done_at, control_bytes, parked_bytes;
};

static systolic_layout_t systolic_layout(int affine, u32 queries_count, u32 candidates_count, u32 longest_query,
                                         u32 longest_candidate) {
    systolic_layout_t layout;
    layout.pairs = (u64)queries_count * candidates_count;
    layout.max_bands = longest_query ? (longest_query + systolic_band_rows_k - 1) / systolic_band_rows_k : 1;
    layout.tickets = layout.pairs * layout.max_bands;
    layout.parked_columns = longest_candidate + 1 + systolic_slack_k;
    layout.progress_at = systolic_header_bytes_k;
    layout.best_at = layout.progress_at + layout.tickets * sizeof(u64);
    layout.done_at = layout.best_at + layout.pairs * sizeof(u64);
    layout.control_bytes = layout.done_at + layout.pairs * sizeof(u64);
    layout.parked_bytes = layout.pairs * (affine ? 2 : 1) * layout.parked_columns * sizeof(i32);
    return layout;
}

constexpr u64 systolic_ticket_limit_k = (1ull << 26) - 16; // one wavefront per ticket: 64 x tickets threads must stay below 2^32

template <bool local_, bool affine_, bool uniform_, bool runes_ = false, bool saturating_ = false>
static int launch_systolic(szs_cost_model_t const *model, szs_string_ref_t const *queries, u32 queries_count,
                           szs_string_ref_t const *candidates, u32 candidates_count, u32 longest_query,
                           u32 longest_candidate, i64 *results, u64 stride, int symmetric, void *control, void *parked,
                           u32 epoch, hipStream_t stream) {
    systolic_layout_t const layout = systolic_layout(affine_, queries_count, candidates_count, longest_query, longest_candidate);
    if (layout.tickets > systolic_ticket_limit_k) return (int)hipErrorInvalidValue; // the host keeps larger jobs on weighted.hip
    char *const base = static_cast<char *>(control);
    u32 const grid = systolic_grid(layout.tickets);
    hipLaunchKernelGGL((systolic_scores_kernel<local_, affine_, uniform_, runes_, saturating_>), dim3(grid),
                       dim3(64 * systolic_waves_k), 0, stream, model, queries, queries_count, candidates, candidates_count,
                       layout.max_bands, results, stride, symmetric, reinterpret_cast<u64 *>(base),
                       reinterpret_cast<u64 *>(base + layout.progress_at), reinterpret_cast<u64 *>(base + layout.best_at),
                       reinterpret_cast<u64 *>(base + layout.done_at), static_cast<i32 *>(parked), layout.parked_columns,
                       epoch);
    return (int)hipGetLastError();
}

} // namespace szs_hip

extern "C" unsigned szs_hip_systolic_band_rows(void) { return szs_hip::systolic_band_rows_k; }

extern "C" int szs_hip_systolic_workspace_bytes(int affine, uint32_t queries_count, uint32_t candidates_count,
                                                uint32_t longest_query, uint32_t longest_candidate, size_t *control_bytes,
                                                size_t *parked_bytes) {
    szs_hip::systolic_layout_t const layout =
        szs_hip::systolic_layout(affine, queries_count, candidates_count, longest_query, longest_candidate);
    if (layout.tickets > szs_hip::systolic_ticket_limit_k) return 0;
    *control_bytes = layout.control_bytes, *parked_bytes = layout.parked_bytes + 256;
    return 1;
}

extern "C" int szs_hip_systolic_scores(int objective, int affine, szs_cost_model_t const *model,
                                       szs_string_ref_t const *queries, uint32_t queries_count,
                                       szs_string_ref_t const *candidates, uint32_t candidates_count,
                                       uint32_t longest_query, uint32_t longest_candidate, int64_t *results,
                                       uint64_t results_row_stride, int symmetric, void *control, void *parked,
                                       uint32_t epoch, void *stream) {
    using namespace szs_hip;
    if (!queries_count || !candidates_count) return 0;
    hipStream_t const s = static_cast<hipStream_t>(stream);
#define SZS_SYSTOLIC_LAUNCH(...)                                                                                       \
    return launch_systolic<__VA_ARGS__>(model, queries, queries_count, candidates, candidates_count, longest_query,    \
                                        longest_candidate, results, results_row_stride, symmetric, control, parked, epoch, s)
    switch (objective) {
    case szs_objective_global_k:
        if (affine) SZS_SYSTOLIC_LAUNCH(false, true, false);
        SZS_SYSTOLIC_LAUNCH(false, false, false);
    case szs_objective_local_k:
        if (affine) SZS_SYSTOLIC_LAUNCH(true, true, false);
        SZS_SYSTOLIC_LAUNCH(true, false, false);
    case szs_objective_local_saturating_k:
        if (affine) SZS_SYSTOLIC_LAUNCH(true, true, false, false, true);
        SZS_SYSTOLIC_LAUNCH(true, false, false, false, true);
    case szs_objective_distance_k:
        if (affine) SZS_SYSTOLIC_LAUNCH(false, true, true);
        SZS_SYSTOLIC_LAUNCH(false, false, true);
    case szs_objective_distance_runes_k:
        if (affine) SZS_SYSTOLIC_LAUNCH(false, true, true, true);
        SZS_SYSTOLIC_LAUNCH(false, false, true, true);
    default: break;
    }
#undef SZS_SYSTOLIC_LAUNCH
    return (int)hipErrorInvalidValue;
}
