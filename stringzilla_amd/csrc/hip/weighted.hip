/*
 *  weighted.hip - Needleman-Wunsch, Smith-Waterman and non-unit-cost Levenshtein scores on gfx950, linear and
 *  affine (Gotoh) gaps, class-table or uniform substitution costs.
 *
 *  Replaces, for the ROCm build, the reference's weighted tiers
 *      weighted_needleman/gotoh_per_cuda_thread_, (affine_)score_per_cuda_warp_, (affine_)score_across_cuda_device_
 *      /root/reference/include/stringzillas/similarities/cuda.cuh:729-1500,5386-5760
 *  and must return exactly what the reference's serial scorers return
 *      needleman_wunsch_score / smith_waterman_score / levenshtein_distance
 *      .../similarities/serial.hpp:2527-2693,2910-3124   (recurrences: tile_scorer, serial.hpp:778-1278).
 *
 *  MI355X-first design - inter-sequence, register-striped, no cross-lane traffic:
 *
 *  - One workgroup = one QUERY x 256 CANDIDATES, one (query, candidate) pair per lane.  Lanes never talk to each
 *    other: there is no anti-diagonal, no shuffle, no shared DP row.  Utilisation comes from the batch (the C-ABI is
 *    a cross-product), and candidates are length-sorted so a wavefront's 64 texts end together.
 *  - The DP matrix of a pair is walked in horizontal STRIPS of 32 query rows.  Inside a strip the lane sweeps the
 *    candidate left to right; the strip's column of 32 cells (plus the gap tracks) lives in VGPRs.  Only the strip's
 *    bottom row must survive until the next strip: it is parked in a per-lane BOUNDARY array in global memory laid out
 *    [column][lane], so a wavefront reads/writes one coalesced 256-byte line per column: 8 B (linear) or 16 B (affine)
 *    of traffic per 32 cells, prefetched one text dword (4 columns) ahead.
 *  - Substitution costs come from a per-strip QUERY PROFILE in LDS indexed by the raw candidate BYTE:
 *    profile[byte][row] = cost(query[row], byte) as packed int8, 32 bytes per symbol, so two ds_read_b128 hand a lane
 *    the costs of 32 cells and the byte -> class map never appears in the inner loop.  The query is shared by the whole
 *    workgroup, so the profile is built once per strip by 256 threads (one symbol each) from an LDS copy of the table.
 *  - Arithmetic is shaped for the measured gfx950 integer rates (scripts/valu_peak.hip: v_add_u32 ~2.6 cycles per
 *    wave-instruction, v_max_i32 / v_max3_i32 / SDWA adds ~4.2):
 *        linear:  sub = diag + cost (one SDWA add, sign-extending the cost byte in place),
 *                 h   = max3(above + gap, left + gap, sub)   where `x + gap` is computed once per produced cell
 *                 -> 3 VALU per cell;
 *        affine:  across = max(left_h + open, across + extend), down = max(above_h + open, down + extend),
 *                 h = max3(down, across, sub) with `h + open`, `across + extend`, `down + extend` each formed once
 *                 -> 7 VALU per cell;
 *        local:   +1 (the substitution branch is clamped at 0) and a max3 tree per column for the running best.
 *  - No per-column length checks in the main loop: it runs while ALL 64 lanes of the wavefront still have a whole text
 *    dword left; the ragged rest (from the wavefront's shortest text to its longest) predicates every column on the
 *    lane's own length, i.e. finished lanes are frozen by the EXEC mask and keep their final column in registers.
 *  - Cells are 32-bit.  The host refuses inputs whose worst-case reach (serial.hpp:135-162) leaves int32.
 *  - Persistent grid sized to what the device can keep RESIDENT (occupancy query x CUs); workgroups pull
 *    (query, candidate-block) items, heaviest first, from one atomic counter, so the boundary workspace is sized by
 *    resident workgroups - not by the results matrix - and ragged batches balance themselves.
 *
 *  Exact boundary values (parity traps of SURVEY.md section 8a) are spelled out next to the code that uses them.
 */
#include "device_common.hpp"

#include <type_traits>

namespace szs_hip {

#ifndef SZS_WEIGHTED_ROWS
#define SZS_WEIGHTED_ROWS 32
#endif
constexpr int weighted_rows_k = SZS_WEIGHTED_ROWS;        // strip height: 32 int8 costs = two ds_read_b128
constexpr u32 weighted_block_threads_k = 256;
constexpr u32 weighted_boundary_slack_k = 8;             // columns the boundary prefetch may run past the longest text

__device__ __forceinline__ i32 max2(i32 a, i32 b) { return a > b ? a : b; }
__device__ __forceinline__ i32 max3(i32 a, i32 b, i32 c) { return max2(max2(a, b), c); }

/** The 32 substitution costs of one text byte against the 32 query rows of the strip, as fetched from LDS. */
struct cost_column_t {
    u32 packed[weighted_rows_k / 4];
    __device__ __forceinline__ i32 operator[](int row) const { return (i32)(int8_t)(packed[row / 4] >> (8 * (row % 4))); }
};

__device__ __forceinline__ cost_column_t load_costs(int8_t const *profile, u32 symbol) {
    uint4 const *rows = reinterpret_cast<uint4 const *>(profile) + symbol * (weighted_rows_k / 16);
    cost_column_t costs;
#pragma unroll
    for (int chunk = 0; chunk < weighted_rows_k / 16; ++chunk) {
        uint4 const part = rows[chunk];
        costs.packed[4 * chunk + 0] = part.x, costs.packed[4 * chunk + 1] = part.y;
        costs.packed[4 * chunk + 2] = part.z, costs.packed[4 * chunk + 3] = part.w;
    }
    return costs;
}

/**
 *  The cells of one strip column, per lane, in the form the recurrences consume them.
 *  Linear gaps keep `h` and `h + gap`; affine gaps keep `h`, `h + open` and the horizontal-gap track pre-extended.
 */
#ifndef SZS_WEIGHTED_RECOMPUTE
#define SZS_WEIGHTED_RECOMPUTE 0
#endif
template <bool affine_>
struct strip_column_t {
    static constexpr bool keeps_gapped = !(affine_ && SZS_WEIGHTED_RECOMPUTE);
    i32 h[weighted_rows_k];            // H(row, column): becomes the diagonal of the next column
    i32 h_gapped[keeps_gapped ? weighted_rows_k : 1]; // H + gap (linear) or H + open (affine): the "left" input of the next column
    i32 across_extended[affine_ ? weighted_rows_k : 1]; // affine: horizontal-gap track + extend
};

/**
 *  `value + gap` in the arithmetic of the kernel instance.
 *  Plain: a signed add.  Saturating (local alignment with non-positive gap costs): every quantity a gap is applied to is
 *  >= 0 - H because the substitution branch is clamped at 0, the two gap tracks because clamping THEM at 0 as well can
 *  never change an H (a negative track value only ever loses against the clamped substitution branch, and what it
 *  propagates stays negative) - so `max(value + gap, 0)` is ONE unsigned saturating subtract of the penalty, a
 *  fast-class VALU op (`v_sub_u32 ... clamp`), and the explicit clamp of the substitution branch disappears:
 *  max3 over two non-negative operands and `diag + cost` is already >= 0.
 */
template <bool saturating_>
__device__ __forceinline__ i32 gapped(i32 value, i32 gap) {
    if constexpr (saturating_) return (i32)__builtin_elementwise_sub_sat((u32)value, (u32)(-gap));
    else return value + gap;
}

/**
 *  Advances one lane by one column of the strip.
 *
 *  @param above_h      H(first_row - 1, j): the row above the strip at this column (border or parked boundary).
 *  @param above_down   affine: the vertical-gap track of that row; ignored for linear gaps.
 *  @param diagonal     H(first_row - 1, j - 1); replaced by `above_h` for the next column.
 *  @param down_out     affine: the vertical-gap track of the strip's bottom row at this column.
 *  @param best         local: running maximum over the rows `[0, counted_rows)` of this column.
 */
template <bool local_, bool affine_, bool saturating_>
__device__ __forceinline__ void advance_column(strip_column_t<affine_> &column, cost_column_t const &costs, i32 above_h,
                                               i32 above_down, i32 &diagonal, i32 gap_open, i32 gap_extend,
                                               i32 &down_out, i32 &best, u32 counted_rows) {
    constexpr int rows = weighted_rows_k;
    i32 diag = diagonal;
    diagonal = above_h;
    i32 above_gapped = gapped<saturating_>(above_h, gap_open);                        // above + gap (linear) / + open
    i32 down_extended = affine_ ? gapped<saturating_>(above_down, gap_extend) : 0;    // vertical-gap track + extend
    i32 down = 0;
#pragma unroll
    for (int r = 0; r < rows; ++r) {
        i32 substituted = diag + costs[r];
        if constexpr (local_ && !saturating_) substituted = max2(substituted, 0); // only this branch is clamped (serial.hpp:957-965)
        diag = column.h[r];
        i32 cell;
        if constexpr (affine_) {
            i32 const left_gapped = strip_column_t<affine_>::keeps_gapped ? column.h_gapped[strip_column_t<affine_>::keeps_gapped ? r : 0]
                                                                          : gapped<saturating_>(diag, gap_open);
            i32 const across = max2(left_gapped, column.across_extended[r]); // serial.hpp:1091-1102
            down = max2(above_gapped, down_extended);
            cell = max3(down, across, substituted);
            column.across_extended[r] = gapped<saturating_>(across, gap_extend);
            down_extended = gapped<saturating_>(down, gap_extend);
        }
        else { cell = max3(above_gapped, column.h_gapped[r], substituted); } // serial.hpp:846-848
        column.h[r] = cell;
        above_gapped = gapped<saturating_>(cell, gap_open);
        if constexpr (strip_column_t<affine_>::keeps_gapped) column.h_gapped[r] = above_gapped;
    }
    down_out = down;
    if constexpr (local_) {
        // Saturating form (gaps <= 0, everything clamped at 0): a padded row - cost 0 against every symbol - can only
        // hold what it inherited, minus penalties, from real cells above it or to its left, all of them counted already.
        // So padded rows may be counted too, and the column needs no branch.
        if (saturating_ || counted_rows >= (u32)rows) { // every row counts: a max3 tree, half an instruction per cell
#pragma unroll
            for (int r = 0; r < rows; r += 2) best = max3(best, column.h[r], column.h[r + 1]);
        }
        else { // last strip of a query whose length is not a multiple of the strip height: padded rows never count
#pragma unroll
            for (int r = 0; r < rows; ++r)
                if ((u32)r < counted_rows) best = max2(best, column.h[r]);
        }
    }
}

/**
 *  @tparam local_    Smith-Waterman (best cell, substitution branch clamped at 0) instead of a global alignment.
 *  @tparam affine_   Gotoh's three-track recurrence instead of the single-track linear one.
 *  @tparam uniform_  costs are (match, mismatch) on raw bytes - weighted Levenshtein, computed as a maximisation of
 *                    negated costs and negated back on output - instead of the 32x32 class table.
 *  @tparam saturating_ (with local_) both gap costs are <= 0: gap arithmetic is unsigned-saturating, see `gapped`.
 *  @tparam narrow_   the host has proven every parked value fits int16 (reach rule for global scores, shortest side x
 *                    largest cost for saturating local ones): the boundary rows are stored as 16-bit values, halving
 *                    the only HBM traffic of the kernel that scales with the DP matrix.
 *  @tparam runes_    (with uniform_) symbols are UTF-32 codepoints: strings are `u32` arrays produced by utf8.hip, lengths
 *                    count runes, and the strip profile is keyed by the slots of a 64-entry rune table of the strip's own
 *                    (at most 32 distinct) runes instead of by byte value; a rune the strip does not contain probes to
 *                    an empty slot, whose profile row is "mismatch against every row".
 */
#ifndef SZS_WEIGHTED_WAVES
#define SZS_WEIGHTED_WAVES 1
#endif
template <bool local_, bool affine_, bool uniform_, bool runes_ = false, bool saturating_ = false, bool narrow_ = false>
__global__ __launch_bounds__(256, SZS_WEIGHTED_WAVES) void weighted_scores_kernel(
    szs_cost_model_t const *__restrict__ model, szs_string_ref_t const *__restrict__ queries, u32 queries_count,
    szs_string_ref_t const *__restrict__ candidates, u32 candidates_count, u32 candidate_blocks,
    i64 *__restrict__ results, u64 results_row_stride, int symmetric, void *__restrict__ boundary, u32 boundary_columns,
    u32 *__restrict__ work_counter) {

    constexpr int rows = weighted_rows_k;
    __shared__ __attribute__((aligned(16))) int8_t profile[256 * rows]; // [candidate byte][row]
    __shared__ int8_t table[32 * 32];                                   // [query class][candidate class]
    __shared__ u8 class_of_byte[256];
    __shared__ u8 strip_classes[rows];                                  // classes (uniform_: bytes) of the strip's rows
    __shared__ u32 claimed_work;
    constexpr u32 strip_slots = 64, strip_slot_empty = ~0u; // rune table of one strip: load factor <= 1/2
    __shared__ u32 strip_keys[runes_ ? strip_slots : 1];
    __shared__ u32 strip_runes[runes_ ? rows : 1];
    static_assert(!runes_ || uniform_, "codepoint scoring exists for uniform costs only");
    static_assert(!saturating_ || local_, "saturating gap arithmetic is a local-alignment form");
    auto strip_slot_of = [&](u32 rune) -> u32 { // the rune's slot, or the empty slot its probe sequence ends on
        u32 slot = (rune * 2654435761u) >> 26;
        for (;;) {
            u32 const key = strip_keys[slot];
            if (key == rune || key == strip_slot_empty) return slot;
            slot = (slot + 1) & (strip_slots - 1);
        }
    };

    i32 const gap_open = model->gap_open, gap_extend = model->gap_extend;
    if constexpr (!uniform_) {
        class_of_byte[threadIdx.x] = model->byte_to_class[threadIdx.x];
        for (u32 i = threadIdx.x; i < 32 * 32; i += weighted_block_threads_k) table[i] = (int8_t)model->substitution[i];
    }
    i32 const uniform_match = model->uniform_match, uniform_mismatch = model->uniform_mismatch;

    // This workgroup's private boundary rows: [column][lane], one plane for H and one for the vertical-gap track.
    u64 const plane = (u64)boundary_columns * weighted_block_threads_k;
    using parked_t = typename std::conditional<narrow_, int16_t, i32>::type;
    parked_t *const boundary_h =
        static_cast<parked_t *>(boundary) + (u64)blockIdx.x * plane * (affine_ ? 2 : 1) + threadIdx.x;
    parked_t *const boundary_down = boundary_h + plane;
    auto parked = [](parked_t *base, u32 j) -> parked_t & { return base[(u64)j * weighted_block_threads_k]; };

    // Work items are (query, candidate block) pairs, handed out through one device-wide counter: queries arrive longest
    // first and candidate blocks are walked from the longest texts down, so the heaviest items start first and the
    // launch drains on its lightest ones, whatever the grid size and however uneven the lengths are.
    u32 const work_items = queries_count * candidate_blocks;
    for (;;) {
        __syncthreads(); // the previous item's LDS (profile, claimed_work) is no longer in use
        if (threadIdx.x == 0) claimed_work = atomicAdd(work_counter, 1u);
        __syncthreads();
        u32 const work = claimed_work;
        if (work >= work_items) break;
        szs_string_ref_t const query = queries[work % queries_count]; // candidate-block-major, heaviest block first
        u32 const candidate_slot =
            (candidate_blocks - 1 - work / queries_count) * weighted_block_threads_k + threadIdx.x;
        bool live = candidate_slot < candidates_count;
        szs_string_ref_t candidate = {0, 0, 0};
        if (live) candidate = candidates[candidate_slot];
        if ((symmetric & SZS_LAYOUT_SYMMETRIC) && candidate.index > query.index) live = false;
        u32 const text_length = live ? candidate.length : 0;
        u32 const longest_in_wave = wave_max_u32(text_length);
        u32 const shortest_in_wave = ~wave_max_u32(live ? ~text_length : 0u); // over live lanes
        // Lanes without a text (dead, or an empty candidate) stream from the workspace instead: always-valid memory, so
        // the branch-free reads of the main loop need no predicate.  Their symbols are never consumed.
        u64 const safe_address = text_length ? candidate.address : (u64)(uintptr_t)boundary_h;
        text_stream_t text(safe_address, text_length);
        if (!text_length) text.valid_dwords = 1;
        u8 const *const pattern = reinterpret_cast<u8 const *>(query.address);
        u32 const query_length = query.length;

        // Value of DP cell (row `i`, column 0) and (row 0, column `j`): the all-gap borders.
        //   global linear : gap * k                                   (serial.hpp:821-823)
        //   global affine : k ? open + extend * (k - 1) : 0           (serial.hpp:1045-1047)
        //   local         : 0
        auto border = [&](u32 k) -> i32 {
            if constexpr (local_) return 0;
            if constexpr (affine_) return k ? gap_open + gap_extend * (i32)(k - 1) : 0;
            return gap_open * (i32)k;
        };

        // A pair with an empty side never enters the column loop: its score is the border itself (local: 0).
        i32 score = local_ ? 0 : border(query_length ? query_length : text_length);

        for (u32 first_row = 0; first_row < query_length; first_row += rows) {
            u32 const rows_here = query_length - first_row < (u32)rows ? query_length - first_row : (u32)rows;
            bool const is_first_strip = first_row == 0;
            bool const is_last_strip = first_row + rows >= query_length;

            // ---- query profile of this strip: thread t owns candidate byte t (runes: rune-table slot t)
            __syncthreads(); // everyone is done with the previous strip's profile (and the table copies are written)
            if constexpr (runes_) {
                if (threadIdx.x < strip_slots) strip_keys[threadIdx.x] = strip_slot_empty;
                __syncthreads();
                if (threadIdx.x < rows_here) {
                    u32 const rune = reinterpret_cast<u32 const *>(query.address)[first_row + threadIdx.x];
                    strip_runes[threadIdx.x] = rune;
                    u32 slot = (rune * 2654435761u) >> 26;
                    for (;;) {
                        u32 const previous = atomicCAS(&strip_keys[slot], strip_slot_empty, rune);
                        if (previous == strip_slot_empty || previous == rune) break;
                        slot = (slot + 1) & (strip_slots - 1);
                    }
                }
            }
            else if (threadIdx.x < (u32)rows) {
                u8 symbol = 0;
                if (threadIdx.x < rows_here) symbol = pattern[first_row + threadIdx.x];
                strip_classes[threadIdx.x] = uniform_ ? symbol : class_of_byte[symbol];
            }
            __syncthreads();
            if (!runes_ || threadIdx.x < strip_slots) {
                u32 const mine = runes_ ? strip_keys[runes_ ? threadIdx.x % strip_slots : 0]
                                        : (uniform_ ? threadIdx.x : (u32)class_of_byte[threadIdx.x]);
                u32 packed[rows / 4];
#pragma unroll
                for (int r = 0; r < rows; ++r) {
                    i32 cost = 0; // padded rows: never read back (global) / never counted (local)
                    if ((u32)r < rows_here) {
                        if constexpr (runes_) cost = strip_runes[r] == mine ? uniform_match : uniform_mismatch;
                        else if constexpr (uniform_) cost = strip_classes[r] == mine ? uniform_match : uniform_mismatch;
                        else // cost(query, candidate) = table[class(query)][class(candidate)]: the QUERY picks the row
                            cost = table[(u32)strip_classes[r] * 32 + mine];
                    }
                    if (r % 4 == 0) packed[r / 4] = 0;
                    packed[r / 4] |= ((u32)cost & 0xFFu) << (8 * (r % 4));
                }
                uint4 *mine_rows = reinterpret_cast<uint4 *>(profile) + threadIdx.x * (rows / 16);
#pragma unroll
                for (int chunk = 0; chunk < rows / 16; ++chunk)
                    mine_rows[chunk] = make_uint4(packed[4 * chunk], packed[4 * chunk + 1], packed[4 * chunk + 2], packed[4 * chunk + 3]);
            }
            __syncthreads();

            // ---- column 0 of the strip.  Track seeds are the FINITE "discard" values of the reference:
            //      global: border + open + extend (serial.hpp:1049-1056); local: open + extend (serial.hpp:1195-1201).
            strip_column_t<affine_> column;
#pragma unroll
            for (int r = 0; r < rows; ++r) {
                column.h[r] = border(first_row + r + 1);
                if constexpr (strip_column_t<affine_>::keeps_gapped) column.h_gapped[r] = gapped<saturating_>(column.h[r], gap_open);
                if constexpr (affine_) // saturating: the (negative) seed is clamped like every other track value
                    column.across_extended[r] = saturating_ ? 0 : column.h[r] + gap_open + gap_extend + gap_extend;
            }
            i32 diagonal = border(first_row); // DP cell (first_row - 1, column - 1)
            i32 best = 0, down_out = 0;

            // The row above the strip at column j (1-based): the border for the first strip, else the parked boundary.
            auto above_of = [&](u32 j, i32 &above_h, i32 &above_down) {
                if (is_first_strip) {
                    above_h = border(j);
                    above_down = saturating_ ? 0 : above_h + gap_open + gap_extend;
                }
                else {
                    above_h = parked(boundary_h, j);
                    above_down = affine_ ? parked(boundary_down, j) : 0;
                }
            };

            u32 column_index = 0, dword = 0; // columns [0, column_index) are done
            u32 const *const runes = reinterpret_cast<u32 const *>(safe_address);
            auto rune_at = [&](u32 index) -> u32 { return index < text_length ? runes[index] : 0u; };
            u32 raw_low = runes_ ? 0u : text.raw(0);
            // The profile row of the symbol in column `column_index + step`, given this batch of four columns.
            auto profile_row = [&](u32 const (&batch)[4], int step) -> u32 {
                if constexpr (runes_) return strip_slot_of(batch[step]);
                else return (batch[0] >> (8 * step)) & 0xFFu;
            };
            // ---- main loop: whole batches of four columns that EVERY live lane of the wavefront still has.
            // Straight-line on purpose: the boundary cells of the next batch are loaded UNCONDITIONALLY (for the first
            // strip they are garbage and the border is selected when they are consumed), the strip's bottom row is
            // stored unconditionally (the last strip's copy is never read), text reads are index-clamped instead of
            // predicated.  No branch splits the batch, so hipcc can count outstanding loads and stores exactly
            // (`s_waitcnt vmcnt(n)` instead of draining the stores of the previous batch before every batch) and can
            // hoist the LDS cost reads of a column above the arithmetic of the previous one.
            if (shortest_in_wave >= 4 && longest_in_wave) {
                u32 raw_high = runes_ ? 0u : text.raw_clamped(1);
                u32 ahead[4] = {0, 0, 0, 0}; // runes: the next batch, loaded one iteration early
                if constexpr (runes_)
                    for (int step = 0; step < 4; ++step) ahead[step] = runes[step];
                parked_t ahead_h[4], ahead_down[affine_ ? 4 : 1];
#pragma unroll
                for (int step = 0; step < 4; ++step) {
                    ahead_h[step] = parked(boundary_h, 1 + step);
                    if constexpr (affine_) ahead_down[step] = parked(boundary_down, 1 + step);
                }
                for (; column_index + 4 <= shortest_in_wave; column_index += 4, ++dword) {
                    u32 batch[4];
                    if constexpr (runes_) {
                        u32 const last = text_length ? text_length - 1 : 0;
#pragma unroll
                        for (int step = 0; step < 4; ++step) batch[step] = ahead[step];
#pragma unroll
                        for (int step = 0; step < 4; ++step) {
                            u32 const index = column_index + 4 + step;
                            ahead[step] = runes[index < last ? index : last];
                        }
                    }
                    else {
                        batch[0] = text.splice(raw_low, raw_high);
                        raw_low = raw_high;
                        raw_high = text.raw_clamped(dword + 2);
                    }
                    i32 now_h[4], now_down[4];
#pragma unroll
                    for (int step = 0; step < 4; ++step) {
                        i32 const edge = border(column_index + 1 + step);
                        now_h[step] = is_first_strip ? edge : (i32)ahead_h[step];
                        now_down[step] = !affine_          ? 0
                                         : !is_first_strip ? (i32)ahead_down[affine_ ? step : 0]
                                         : saturating_     ? 0
                                                           : edge + gap_open + gap_extend;
                    }
                    // Prefetch the next batch's boundary cells; the slack columns make the overrun harmless.
#pragma unroll
                    for (int step = 0; step < 4; ++step) {
                        ahead_h[step] = parked(boundary_h, column_index + 5 + step);
                        if constexpr (affine_) ahead_down[step] = parked(boundary_down, column_index + 5 + step);
                    }
#pragma unroll
                    for (int step = 0; step < 4; ++step) {
                        cost_column_t const costs = load_costs(profile, profile_row(batch, step));
                        advance_column<local_, affine_, saturating_>(column, costs, now_h[step], now_down[step], diagonal,
                                                                     gap_open, gap_extend, down_out, best, rows_here);
                        parked(boundary_h, column_index + step + 1) = (parked_t)column.h[rows - 1];
                        if constexpr (affine_) parked(boundary_down, column_index + step + 1) = (parked_t)down_out;
                    }
                }
            }
            // ---- ragged rest: every column predicated on this lane's own length
            if (column_index < longest_in_wave) {
                u32 raw_high = runes_ ? 0u : text.raw(dword + 1);
#pragma unroll 1
                for (; column_index < longest_in_wave; column_index += 4, ++dword) {
                    u32 batch[4] = {0, 0, 0, 0};
                    if constexpr (runes_) {
#pragma unroll
                        for (int step = 0; step < 4; ++step) batch[step] = rune_at(column_index + step);
                    }
                    else {
                        batch[0] = text.splice(raw_low, raw_high);
                        raw_low = raw_high;
                        raw_high = text.raw(dword + 2);
                    }
#pragma unroll
                    for (int step = 0; step < 4; ++step) {
                        u32 const j = column_index + step + 1; // 1-based DP column
                        if (j <= text_length) {
                            i32 above_h, above_down;
                            above_of(j, above_h, above_down);
                            cost_column_t const costs = load_costs(profile, profile_row(batch, step));
                            advance_column<local_, affine_, saturating_>(column, costs, above_h, above_down, diagonal, gap_open,
                                                            gap_extend, down_out, best, rows_here);
                            if (!is_last_strip) {
                                parked(boundary_h, j) = (parked_t)column.h[rows - 1];
                                if constexpr (affine_) parked(boundary_down, j) = (parked_t)down_out;
                            }
                        }
                    }
                }
            }

            if constexpr (local_) score = max2(score, best);
            else if (is_last_strip) { // bottom-right cell: last real row of the last strip, frozen at the lane's last column
#pragma unroll
                for (int r = 0; r < rows; ++r)
                    if ((u32)r + 1 == rows_here) score = column.h[r];
            }
        }

        if (live) {
            i64 const value = uniform_ ? -(i64)score : (i64)score;
            bool const transposed = (symmetric & SZS_LAYOUT_TRANSPOSED) != 0; // kernel roles swapped by the host
            u64 const row = transposed ? candidate.index : query.index, column = transposed ? query.index : candidate.index;
            results[row * results_row_stride + column] = value;
            if ((symmetric & SZS_LAYOUT_SYMMETRIC) && candidate.index != query.index)
                results[column * results_row_stride + row] = value;
        }
    }
}

constexpr size_t weighted_header_bytes_k = 256; // the work counter lives at the head of the boundary workspace

/** Workgroups that can be RESIDENT at once for this kernel instance on the current device (never more than the work). */
template <bool local_, bool affine_, bool uniform_, bool runes_ = false, bool saturating_ = false, bool narrow_ = false>
static u32 weighted_grid(u64 work_items) {
    static int resident_of[device_slots_k]; // per instance and device ordinal
    int *const slot = &resident_of[device_slot()];
    int resident = cached(slot);
    if (!resident) {
        int device = 0, units = 0, per_unit = 0;
        if (hipGetDevice(&device) != hipSuccess ||
            hipDeviceGetAttribute(&units, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess ||
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_unit, weighted_scores_kernel<local_, affine_, uniform_, runes_, saturating_, narrow_>,
                                                         (int)weighted_block_threads_k, 0) != hipSuccess ||
            units <= 0 || per_unit <= 0) {
            (void)hipGetLastError();
            units = 256, per_unit = 2;
        }
        resident = units * per_unit;
        remember(slot, resident);
    }
    return (u32)(work_items < (u64)resident ? work_items : (u64)resident);
}

static u64 weighted_work_items(u32 queries_count, u32 candidates_count) {
    return (u64)queries_count * ((candidates_count + weighted_block_threads_k - 1) / weighted_block_threads_k);
}

template <bool local_, bool affine_, bool uniform_, bool runes_ = false, bool saturating_ = false, bool narrow_ = false>
static size_t weighted_workspace_bytes(u32 queries_count, u32 candidates_count, u32 longest_candidate) {
    u32 const grid = weighted_grid<local_, affine_, uniform_, runes_, saturating_, narrow_>(weighted_work_items(queries_count, candidates_count));
    return weighted_header_bytes_k + (size_t)grid * (longest_candidate + 1 + weighted_boundary_slack_k) *
                                         weighted_block_threads_k * (narrow_ ? sizeof(int16_t) : sizeof(i32)) * (affine_ ? 2 : 1);
}

template <bool local_, bool affine_, bool uniform_, bool runes_ = false, bool saturating_ = false, bool narrow_ = false>
static int launch_weighted(szs_cost_model_t const *model, szs_string_ref_t const *queries, u32 queries_count,
                           szs_string_ref_t const *candidates, u32 candidates_count, u32 longest_candidate, i64 *results,
                           u64 stride, int symmetric, void *workspace, hipStream_t stream) {
    u32 const candidate_blocks = (candidates_count + weighted_block_threads_k - 1) / weighted_block_threads_k;
    u64 const work_items = weighted_work_items(queries_count, candidates_count);
    if (work_items > 0xFFFFFFF0ull) return (int)hipErrorInvalidValue; // the host cuts larger cross-products
    u32 const grid = weighted_grid<local_, affine_, uniform_, runes_, saturating_, narrow_>(work_items);
    u32 *const counter = static_cast<u32 *>(workspace);
    void *const boundary = static_cast<char *>(workspace) + weighted_header_bytes_k;
    hipError_t error = hipMemsetAsync(counter, 0, sizeof(u32), stream);
    if (error != hipSuccess) return (int)error;
    hipLaunchKernelGGL((weighted_scores_kernel<local_, affine_, uniform_, runes_, saturating_, narrow_>), dim3(grid), dim3(weighted_block_threads_k), 0,
                       stream, model, queries, queries_count, candidates, candidates_count, candidate_blocks, results,
                       stride, symmetric, boundary, longest_candidate + 1 + weighted_boundary_slack_k, counter);
    return (int)hipGetLastError();
}

} // namespace szs_hip

#define SZS_WEIGHTED_DISPATCH(CALL)                                                                                   \
    switch (objective) {                                                                                               \
    case szs_objective_global_k:                                                                                       \
        if (narrow && affine) CALL(false, true, false, false, false, true);                                            \
        if (narrow) CALL(false, false, false, false, false, true);                                                     \
        if (affine) CALL(false, true, false);                                                                          \
        CALL(false, false, false);                                                                                     \
    case szs_objective_local_k:                                                                                        \
        if (affine) CALL(true, true, false);                                                                           \
        CALL(true, false, false);                                                                                      \
    case szs_objective_local_saturating_k:                                                                             \
        if (narrow && affine) CALL(true, true, false, false, true, true);                                              \
        if (narrow) CALL(true, false, false, false, true, true);                                                       \
        if (affine) CALL(true, true, false, false, true);                                                              \
        CALL(true, false, false, false, true);                                                                         \
    case szs_objective_distance_k:                                                                                     \
        if (affine) CALL(false, true, true);                                                                           \
        CALL(false, false, true);                                                                                      \
    case szs_objective_distance_runes_k:                                                                               \
        if (affine) CALL(false, true, true, true);                                                                     \
        CALL(false, false, true, true);                                                                                \
    default: break;                                                                                                    \
    }

extern "C" size_t szs_hip_weighted_boundary_bytes(int objective, int affine, int narrow, uint32_t queries_count,
                                                  uint32_t candidates_count, uint32_t longest_candidate) {
    using namespace szs_hip;
#define SZS_WEIGHTED_BYTES(...)                                                                                       \
    return weighted_workspace_bytes<__VA_ARGS__>(queries_count, candidates_count, longest_candidate)
    SZS_WEIGHTED_DISPATCH(SZS_WEIGHTED_BYTES)
#undef SZS_WEIGHTED_BYTES
    return 0;
}

extern "C" int szs_hip_weighted_scores(int objective, int affine, int narrow, szs_cost_model_t const *model,
                                       szs_string_ref_t const *queries, uint32_t queries_count,
                                       szs_string_ref_t const *candidates, uint32_t candidates_count,
                                       uint32_t longest_candidate, int64_t *results, uint64_t results_row_stride,
                                       int symmetric, void *boundary, void *stream) {
    using namespace szs_hip;
    if (!queries_count || !candidates_count) return 0;
    hipStream_t const s = static_cast<hipStream_t>(stream);
#define SZS_WEIGHTED_LAUNCH(...)                                                                                      \
    return launch_weighted<__VA_ARGS__>(model, queries, queries_count, candidates, candidates_count, longest_candidate, \
                                        results, results_row_stride, symmetric, boundary, s)
    SZS_WEIGHTED_DISPATCH(SZS_WEIGHTED_LAUNCH)
#undef SZS_WEIGHTED_LAUNCH
    return (int)hipErrorInvalidValue;
}
