/*
 *  weighted_packed.hip - Needleman-Wunsch and Smith-Waterman scores over a class table (BLOSUM62, NUC.4.4, custom
 *  32 x 32) when every DP value provably fits 16 bits: the lanes tier of weighted.hip with TWO cells per VALU operation.
 *
 *  Same contract and same data flow as weighted_scores_kernel (one query x 256 candidates per workgroup, one pair per
 *  lane, strips of 32 query rows in registers, strip bottom rows parked [column][lane] in global memory); results must
 *  equal the reference's serial scorers bit for bit
 *      needleman_wunsch_score / smith_waterman_score   .../similarities/serial.hpp:2910-3124
 *      recurrences: tile_scorer                        .../similarities/serial.hpp:778-1278
 *  whose own SIMD tiers narrow their cells by the same reach rule (serial.hpp:135-162,370-386).
 *
 *  What is different - the integer VALU of gfx950 retires `v_pk_add_i16` / `v_pk_max_i16` / `v_pk_sub_u16 clamp` at the
 *  rate of their 32-bit twins (scripts/valu_peak.hip), and weighted.hip is bound by exactly that rate (PMC: 8.2 VALU
 *  lane-operations per affine cell, VALU busy 81 %, profiles/r01/pmc_cfg4_v1.json):
 *
 *  - A strip is split into an UPPER half A (rows 0-15) and a LOWER half B (rows 16-31).  Register r holds row r of A in
 *    its low 16 bits and row r of B in its high 16 bits.  B runs ONE COLUMN BEHIND A: at step t, A scores column t and B
 *    scores column t - 1, so the row B needs from above - A's last row at column t - 1 - is what A produced one step
 *    earlier and already sits in the low half of register 15: one `v_lshl_or_b32` per step hands it over.  The two
 *    halves of a register are never the same column, so no operation ever needs a cross-half dependency.
 *  - Substitution costs of both halves arrive with ONE LDS read per register: the strip profile is keyed by the PAIR
 *    (class of text symbol t, class of text symbol t - 1) and holds (cost(A row r, symbol t), cost(B row r, symbol t - 1))
 *    as packed int16: (classes + 1)^2 x 16 dwords, 40 KB for BLOSUM62, 18 KB for NUC.4.4 (dynamic LDS).  The extra class
 *    is the NULL symbol (cost 0 against every row) that A is fed on its one step past the end of the text.
 *  - Linear gaps: 4 packed operations per 2 cells (weighted.hip: 3 per cell); affine: 8 per 2 cells (7 per cell).
 *
 *  Edges, all exact:
 *    step 1     B would score "column 0"; its half is recomputed from the border formulas right after (global only: with
 *               saturating local arithmetic an all-zero column stays all-zero);
 *    step n + 1 only B has a column left.  It runs once, after the column loop, for all lanes at their own n; A's half
 *               scores the null symbol and is never looked at again (global: A's cell was captured before; local: with
 *               gaps <= 0 and clamping at 0 a cost-0 column can only hold what real cells already reached - the
 *               argument that lets padded rows be counted in weighted.hip);
 *    ragged     lanes are frozen by EXEC after their own step n, state and all, exactly as in weighted.hip.
 *
 *  The host takes this kernel when the engine is a class-table one, gaps of a local engine are <= 0, and the reach
 *  (global) / shortest side x largest cost (local) stays below 32000; `SZS_ROCM_PACKED=0` pins the 32-bit kernel.
 */
#include "device_common.hpp"

#include <cstdlib>
#include <type_traits>

namespace szs_hip {

typedef short pk_i16 __attribute__((ext_vector_type(2)));
typedef unsigned short pk_u16 __attribute__((ext_vector_type(2)));

constexpr int packed_registers_k = 16;                   // registers per strip track: 32 query rows
constexpr int packed_strip_rows_k = 2 * packed_registers_k;
#ifndef SZS_PACKED_BLOCK_THREADS
#define SZS_PACKED_BLOCK_THREADS 256
#endif
constexpr u32 packed_block_threads_k = SZS_PACKED_BLOCK_THREADS;
constexpr u32 packed_boundary_slack_k = 8;               // columns the boundary prefetch may run past the longest text
constexpr size_t packed_header_bytes_k = 256;            // the work counter lives at the head of the boundary workspace

__device__ __forceinline__ pk_i16 pk_bits(u32 bits) { return __builtin_bit_cast(pk_i16, bits); }
__device__ __forceinline__ u32 pk_raw(pk_i16 value) { return __builtin_bit_cast(u32, value); }
__device__ __forceinline__ pk_i16 pk_pair(i32 low, i32 high) { return pk_bits(((u32)low & 0xFFFFu) | ((u32)high << 16)); }
__device__ __forceinline__ pk_i16 pk_max(pk_i16 a, pk_i16 b) { return __builtin_elementwise_max(a, b); }
__device__ __forceinline__ i32 pk_low(pk_i16 value) { return (i32)value.x; }
__device__ __forceinline__ i32 pk_high(pk_i16 value) { return (i32)value.y; }
__device__ __forceinline__ i32 packed_max_i32(i32 a, i32 b) { return a > b ? a : b; }

/** `value + gap` for both halves; `gap` holds the signed cost twice, or - saturating form - the penalty (-cost) twice. */
template <bool saturating_>
__device__ __forceinline__ pk_i16 pk_gapped(pk_i16 value, pk_i16 gap) {
    if constexpr (saturating_)
        return __builtin_bit_cast(pk_i16, __builtin_elementwise_sub_sat(__builtin_bit_cast(pk_u16, value), __builtin_bit_cast(pk_u16, gap)));
    else return value + gap;
}

template <bool affine_>
struct packed_column_t {
    pk_i16 h[packed_registers_k];                          // H: A at its current column, B one column behind
    pk_i16 h_gapped[packed_registers_k];                   // H + gap (linear) / H + open (affine)
    pk_i16 across_extended[affine_ ? packed_registers_k : 1]; // affine: horizontal-gap track + extend
};

/** The 16 packed cost pairs of one (symbol, previous symbol) class pair. */
struct packed_costs_t {
    u32 pairs[packed_registers_k];
};

/** Two layouts of the pair profile.
 *  chunk-major  [chunk of 4 registers][class pair][4 registers]: within one ds_read_b128 the 64 lanes' addresses differ by
 *               multiples of 16 B and spread over all bank groups.  Round 1 kept a pair's 16 registers together (64-byte
 *               rows): every lane of an instruction then hit one of only two (four) bank groups and the LDS was busy 89 % of
 *               config 3's kernel time, 72 % of it in conflicts (profiles/r02/pmc_configs.json); chunk-major: 25.0 -> 23.0 ms.
 *  pair-major   [class pair][16 registers]: one address, four immediate offsets - three VALU additions fewer per step.  The
 *               affine kernels do twice the arithmetic per LDS byte and are bound by VALU issue (LDS busy 29 %): they keep it
 *               (config 4: 793 ms pair-major, 809 ms chunk-major). */
template <bool chunk_major_>
__device__ __forceinline__ packed_costs_t load_packed_costs(u32 const *profile, u32 class_pair, u32 class_pairs) {
    uint4 const *rows = reinterpret_cast<uint4 const *>(profile) + (chunk_major_ ? class_pair : class_pair * (packed_registers_k / 4));
    packed_costs_t costs;
#pragma unroll
    for (int chunk = 0; chunk < packed_registers_k / 4; ++chunk) {
        uint4 const part = rows[chunk_major_ ? chunk * class_pairs : chunk];
        costs.pairs[4 * chunk + 0] = part.x, costs.pairs[4 * chunk + 1] = part.y;
        costs.pairs[4 * chunk + 2] = part.z, costs.pairs[4 * chunk + 3] = part.w;
    }
    return costs;
}

/**
 *  One step: A advances to its next column, B to the column A has just left.
 *  `above_h` / `above_down`: low half = the row above the strip at A's column, high half = A's last row at B's column.
 */
template <bool local_, bool affine_>
__device__ __forceinline__ void packed_advance(packed_column_t<affine_> &column, packed_costs_t const &costs, pk_i16 above_h,
                                               pk_i16 above_down, pk_i16 &diagonal, pk_i16 gap_open, pk_i16 gap_extend,
                                               pk_i16 &down_out, pk_i16 (&best)[4]) {
    constexpr bool saturating_ = local_;
    pk_i16 diag = diagonal;
    diagonal = above_h;
    pk_i16 above_gapped = pk_gapped<saturating_>(above_h, gap_open);
    pk_i16 down_extended = affine_ ? pk_gapped<saturating_>(above_down, gap_extend) : pk_bits(0);
    pk_i16 down = pk_bits(0);
#pragma unroll
    for (int r = 0; r < packed_registers_k; ++r) {
        pk_i16 const substituted = diag + pk_bits(costs.pairs[r]); // local: >= -128, and the other two branches are >= 0
        diag = column.h[r];
        pk_i16 cell;
        if constexpr (affine_) {
            pk_i16 const across = pk_max(column.h_gapped[r], column.across_extended[r]); // serial.hpp:1091-1102
            down = pk_max(above_gapped, down_extended);
            cell = pk_max(down, pk_max(across, substituted)); // the inner maximum does not wait for the row above
            column.across_extended[r] = pk_gapped<saturating_>(across, gap_extend);
            down_extended = pk_gapped<saturating_>(down, gap_extend);
        }
        else { cell = pk_max(above_gapped, pk_max(column.h_gapped[r], substituted)); } // serial.hpp:846-848
        column.h[r] = cell;
        above_gapped = pk_gapped<saturating_>(cell, gap_open);
        column.h_gapped[r] = above_gapped;
        // four running maxima: back-to-back DEPENDENT packed operations cost a wait state each on gfx950 (hipcc pads them
        // with s_nop), independent ones do not
        if constexpr (local_) best[r % 4] = pk_max(best[r % 4], cell);
    }
    down_out = down;
}

/**
 *  @tparam local_   Smith-Waterman with gap costs <= 0 (unsigned-saturating gap arithmetic, see weighted.hip `gapped`)
 *                   instead of Needleman-Wunsch.
 *  @tparam affine_  Gotoh's three-track recurrence instead of the single-track linear one.
 */
#ifndef SZS_PACKED_AFFINE_WAVES
#define SZS_PACKED_AFFINE_WAVES 3
#endif
template <bool local_, bool affine_>
__global__ __launch_bounds__(packed_block_threads_k, affine_ ? SZS_PACKED_AFFINE_WAVES : 4) void weighted_packed_kernel( // three / four wavefronts per SIMD: no spills either way
    szs_cost_model_t const *__restrict__ model, szs_string_ref_t const *__restrict__ queries, u32 queries_count,
    szs_string_ref_t const *__restrict__ candidates, u32 candidates_count, u32 candidate_blocks,
    i64 *__restrict__ results, u64 results_row_stride, int layout_flags, int16_t *__restrict__ boundary,
    u32 boundary_columns, u32 *__restrict__ work_counter, u32 classes) {

    constexpr bool saturating_ = local_;
    constexpr int registers = packed_registers_k, rows = packed_strip_rows_k;
    extern __shared__ __attribute__((aligned(16))) u32 pair_profile[]; // [register chunk][class of symbol t * K + class of symbol t - 1][4]
    // The 32 x 32 table itself stays in global memory (2 KB, cache-resident, read only while a strip's profile is built):
    // BLOSUM62's 25 x 25 pair profile is 40,000 B, and with a kilobyte less of static LDS FOUR workgroups fit a CU instead
    // of three - config 3's 4096 work items then take 4 full rounds instead of 5.33 (profiles/r02).
    int16_t const *const table = model->substitution;                   // [query class][candidate class]
    __shared__ u8 class_of_byte[256];
    __shared__ u8 strip_classes[rows];                                 // 0xFF: a padded row
    __shared__ u32 claimed_work;

    u32 const class_slots = classes + 1, null_class = classes; // the null symbol costs 0 against every row
    constexpr bool chunk_major = !affine_;                      // layout of the pair profile: see load_packed_costs
    i32 const gap_open = model->gap_open, gap_extend = model->gap_extend;
    pk_i16 const open_pk = saturating_ ? pk_pair(-gap_open, -gap_open) : pk_pair(gap_open, gap_open);
    pk_i16 const extend_pk = saturating_ ? pk_pair(-gap_extend, -gap_extend) : pk_pair(gap_extend, gap_extend);
    for (u32 byte = threadIdx.x; byte < 256; byte += packed_block_threads_k) class_of_byte[byte] = model->byte_to_class[byte];

    // This workgroup's private boundary rows: [column][lane], one plane for H and one for the vertical-gap track.
    u64 const plane = (u64)boundary_columns * packed_block_threads_k;
    int16_t *const boundary_h = boundary + (u64)blockIdx.x * plane * (affine_ ? 2 : 1) + threadIdx.x;
    int16_t *const boundary_down = boundary_h + plane;
    auto parked = [](int16_t *base, u32 j) -> int16_t & { return base[(u64)j * packed_block_threads_k]; };

    u32 const work_items = queries_count * candidate_blocks;
    for (;;) {
        __syncthreads(); // the previous item's LDS (profile, claimed_work) is no longer in use
        if (threadIdx.x == 0) claimed_work = atomicAdd(work_counter, 1u);
        __syncthreads();
        u32 const work = claimed_work;
        if (work >= work_items) break;
        // candidate-block-major, heaviest block first (lev_myers.hip: myers_work_item): +7 % on config 3, +6 % on config 4
        szs_string_ref_t const query = queries[work % queries_count];
        u32 const candidate_slot = (candidate_blocks - 1 - work / queries_count) * packed_block_threads_k + threadIdx.x;
        bool live = candidate_slot < candidates_count;
        szs_string_ref_t candidate = {0, 0, 0};
        if (live) candidate = candidates[candidate_slot];
        if ((layout_flags & SZS_LAYOUT_SYMMETRIC) && candidate.index > query.index) live = false;
        u32 const text_length = live ? candidate.length : 0;
        u32 const longest_in_wave = wave_max_u32(text_length);
        u32 const shortest_in_wave = ~wave_max_u32(live ? ~text_length : 0u); // over live lanes
        u64 const safe_address = text_length ? candidate.address : (u64)(uintptr_t)boundary_h; // see weighted.hip
        text_stream_t text(safe_address, text_length);
        if (!text_length) text.valid_dwords = 1;
        u8 const *const pattern = reinterpret_cast<u8 const *>(query.address);
        u32 const query_length = query.length;

        // DP cell (row i, column 0) and (row 0, column j): the all-gap borders (serial.hpp:821-823,1045-1047); local: 0.
        auto border = [&](u32 k) -> i32 {
            if constexpr (local_) return 0;
            if constexpr (affine_) return k ? gap_open + gap_extend * (i32)(k - 1) : 0;
            return gap_open * (i32)k;
        };
        i32 score = local_ ? 0 : border(query_length ? query_length : text_length); // an empty side never enters the loop

        for (u32 first_row = 0; first_row < query_length; first_row += rows) {
            u32 const rows_here = query_length - first_row < (u32)rows ? query_length - first_row : (u32)rows;
            bool const is_first_strip = first_row == 0;
            bool const is_last_strip = first_row + rows >= query_length;

            // ---- pair profile of this strip: thread p owns class pairs p, p + 256, ...
            __syncthreads(); // everyone is done with the previous strip's profile (and the table copies are written)
            if (threadIdx.x < (u32)rows)
                strip_classes[threadIdx.x] = threadIdx.x < rows_here ? class_of_byte[pattern[first_row + threadIdx.x]] : (u8)0xFF;
            __syncthreads();
            for (u32 pair = threadIdx.x; pair < class_slots * class_slots; pair += packed_block_threads_k) {
                u32 const upper_class = pair / class_slots, lower_class = pair % class_slots; // A: symbol t, B: symbol t - 1
                u32 packed[registers];
#pragma unroll
                for (int r = 0; r < registers; ++r) {
                    u32 const upper_row = strip_classes[r], lower_row = strip_classes[registers + r];
                    // cost(query, candidate) = table[class(query)][class(candidate)]: the QUERY picks the row
                    i32 const upper = upper_row != 0xFF && upper_class != null_class ? table[upper_row * 32 + upper_class] : 0;
                    i32 const lower = lower_row != 0xFF && lower_class != null_class ? table[lower_row * 32 + lower_class] : 0;
                    packed[r] = ((u32)upper & 0xFFFFu) | ((u32)lower << 16);
                }
                uint4 *const mine = reinterpret_cast<uint4 *>(pair_profile) + (chunk_major ? pair : pair * (registers / 4));
#pragma unroll
                for (int chunk = 0; chunk < registers / 4; ++chunk)
                    mine[chunk_major ? chunk * class_slots * class_slots : chunk] = make_uint4(packed[4 * chunk], packed[4 * chunk + 1], packed[4 * chunk + 2], packed[4 * chunk + 3]);
            }
            __syncthreads();

            // ---- column 0 of the strip.  Track seeds are the FINITE "discard" values of the reference:
            //      global: border + open + extend (serial.hpp:1049-1056); local: clamped like every other track value.
            packed_column_t<affine_> column;
            auto seed_column = [&](int r, bool lower_half_only) {
                i32 const upper = border(first_row + r + 1), lower = border(first_row + registers + r + 1);
                pk_i16 const h = pk_pair(upper, lower), gapped = pk_gapped<saturating_>(h, open_pk);
                pk_i16 const across = saturating_ ? pk_bits(0) : pk_pair(upper + gap_open + 2 * gap_extend, lower + gap_open + 2 * gap_extend);
                u32 const keep = lower_half_only ? 0x0000FFFFu : 0u; // bits that survive from the current value
                column.h[r] = pk_bits((pk_raw(column.h[r]) & keep) | (pk_raw(h) & ~keep));
                column.h_gapped[r] = pk_bits((pk_raw(column.h_gapped[r]) & keep) | (pk_raw(gapped) & ~keep));
                if constexpr (affine_)
                    column.across_extended[r] = pk_bits((pk_raw(column.across_extended[r]) & keep) | (pk_raw(across) & ~keep));
            };
#pragma unroll
            for (int r = 0; r < registers; ++r) {
                column.h[r] = column.h_gapped[r] = pk_bits(0);
                if constexpr (affine_) column.across_extended[r] = pk_bits(0);
                seed_column(r, false);
            }
            pk_i16 diagonal = pk_pair(border(first_row), 0); // A: DP cell (first_row, 0); B: set by the first hand-over
            pk_i16 best[4] = {pk_bits(0), pk_bits(0), pk_bits(0), pk_bits(0)}, down_out = pk_bits(0);
            u32 previous_class = null_class;

            // One step for this lane: A scores DP column `j` (symbol class `upper_class`), B scores column j - 1.
            auto step = [&](u32 upper_class, i32 above_h, i32 above_down) {
                packed_costs_t const costs = load_packed_costs<chunk_major>(pair_profile, upper_class * class_slots + previous_class, class_slots * class_slots);
                previous_class = upper_class;
                // hand-over: A's last row at column j - 1 (low half of register 15, before this step) becomes B's row above
                pk_i16 const above_pk = pk_bits((pk_raw(column.h[registers - 1]) << 16) | ((u32)above_h & 0xFFFFu));
                pk_i16 const above_down_pk = pk_bits((pk_raw(down_out) << 16) | ((u32)above_down & 0xFFFFu));
                packed_advance<local_, affine_>(column, costs, above_pk, above_down_pk, diagonal, open_pk, extend_pk, down_out, best);
            };
            // The strip's bottom row is B's last row: after the step in which A scored column j it is column j - 1.
            auto park = [&](int16_t *slot_h, int16_t *slot_down) {
                *slot_h = (int16_t)pk_high(column.h[registers - 1]);
                if constexpr (affine_) *slot_down = (int16_t)pk_high(down_out);
            };
            // After step 1 B's half holds a scored "column 0": put the borders back (local: it is all zeros anyway).
            auto restore_lower_half = [&]() {
                if constexpr (!local_) {
#pragma unroll
                    for (int r = 0; r < registers; ++r) seed_column(r, true);
                }
            };
            // The row above the strip at column j (1-based): the border for the first strip, else the parked boundary.
            auto above_of = [&](u32 j, i32 &above_h, i32 &above_down) {
                if (is_first_strip) {
                    above_h = border(j);
                    above_down = saturating_ ? 0 : above_h + gap_open + gap_extend;
                }
                else {
                    above_h = parked(boundary_h, j);
                    above_down = affine_ ? (i32)parked(boundary_down, j) : 0;
                }
            };

            u32 column_index = 0, dword = 0; // A has scored the columns [1, column_index]
            u32 raw_low = text.raw(0);
            // ---- main loop: whole batches of four columns that EVERY live lane of the wavefront still has; straight-line
            //      (unconditional boundary loads / stores, index-clamped text reads) for the reasons given in weighted.hip.
            if (shortest_in_wave >= 4 && longest_in_wave) {
                u32 raw_high = text.raw_clamped(1);
                int16_t ahead_h[4], ahead_down[affine_ ? 4 : 1];
                // cursors at column `column_index`: every access of a batch is a constant offset from them
                constexpr u32 pitch = packed_block_threads_k;
                int16_t *cursor_h = boundary_h, *cursor_down = boundary_down;
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    ahead_h[s] = cursor_h[(1 + s) * pitch];
                    if constexpr (affine_) ahead_down[s] = cursor_down[(1 + s) * pitch];
                }
                u32 bytes = text.splice(raw_low, raw_high);
                raw_low = raw_high, raw_high = text.raw_clamped(2);
                u32 classes_now[4];
#pragma unroll
                for (int s = 0; s < 4; ++s) classes_now[s] = class_of_byte[(bytes >> (8 * s)) & 0xFFu];
                auto batch = [&](auto first_batch) {
                    constexpr bool first_ = decltype(first_batch)::value;
                    // the text and the classes of the NEXT batch, one batch early
                    u32 const bytes_ahead = text.splice(raw_low, raw_high);
                    raw_low = raw_high, raw_high = text.raw_clamped(dword + 3);
                    u32 classes_ahead[4];
#pragma unroll
                    for (int s = 0; s < 4; ++s) classes_ahead[s] = class_of_byte[(bytes_ahead >> (8 * s)) & 0xFFu];
                    i32 now_h[4], now_down[4];
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        i32 const edge = border(column_index + 1 + s);
                        now_h[s] = is_first_strip ? edge : (i32)ahead_h[s];
                        now_down[s] = !affine_ ? 0 : !is_first_strip ? (i32)ahead_down[affine_ ? s : 0] : saturating_ ? 0 : edge + gap_open + gap_extend;
                    }
#pragma unroll
                    for (int s = 0; s < 4; ++s) { // the slack columns make the overrun harmless
                        ahead_h[s] = cursor_h[(5 + s) * pitch];
                        if constexpr (affine_) ahead_down[s] = cursor_down[(5 + s) * pitch];
                    }
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        step(classes_now[s], now_h[s], now_down[s]);
                        if (first_ && s == 0) restore_lower_half();
                        else park(cursor_h + s * pitch, cursor_down + s * pitch);
                    }
#pragma unroll
                    for (int s = 0; s < 4; ++s) classes_now[s] = classes_ahead[s];
                    column_index += 4, ++dword, cursor_h += 4 * pitch, cursor_down += 4 * pitch;
                };
                batch(std::true_type {});
                while (column_index + 4 <= shortest_in_wave) batch(std::false_type {});
                raw_low = text.raw(dword); // the ragged rest re-reads its dwords with bounds checks
            }
            // ---- ragged rest: every column predicated on this lane's own length
            if (column_index < longest_in_wave) {
                u32 raw_high = text.raw(dword + 1);
#pragma unroll 1
                for (; column_index < longest_in_wave; column_index += 4, ++dword) {
                    u32 const bytes = text.splice(raw_low, raw_high);
                    raw_low = raw_high;
                    raw_high = text.raw(dword + 2);
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        u32 const j = column_index + s + 1; // 1-based DP column
                        if (j <= text_length) {
                            i32 above_h, above_down;
                            above_of(j, above_h, above_down);
                            step(class_of_byte[(bytes >> (8 * s)) & 0xFFu], above_h, above_down);
                            if (j == 1) restore_lower_half();
                            else if (!is_last_strip) park(&parked(boundary_h, j - 1), &parked(boundary_down, j - 1));
                        }
                    }
                }
            }

            // ---- A is done: its bottom-right candidate is captured before the one step that only B still needs
            u32 const last_row = (query_length - 1) % rows; // of the last strip, if this is it
            if constexpr (!local_) {
                if (is_last_strip && last_row < (u32)registers) {
#pragma unroll
                    for (int r = 0; r < registers; ++r)
                        if ((u32)r == last_row) score = pk_low(column.h[r]);
                }
            }
            if (text_length) { // step n + 1, every lane at its own n: A scores the null symbol, B its last column
                step(null_class, 0, 0);
                if (!is_last_strip) park(&parked(boundary_h, text_length), &parked(boundary_down, text_length));
                if constexpr (local_) {
                    pk_i16 const both = pk_max(pk_max(best[0], best[1]), pk_max(best[2], best[3]));
                    score = packed_max_i32(score, packed_max_i32(pk_low(both), pk_high(both)));
                }
                else if (is_last_strip && last_row >= (u32)registers) {
#pragma unroll
                    for (int r = 0; r < registers; ++r)
                        if ((u32)r + registers == last_row) score = pk_high(column.h[r]);
                }
            }
        }

        if (live) {
            bool const transposed = (layout_flags & SZS_LAYOUT_TRANSPOSED) != 0; // kernel roles swapped by the host
            u64 const row = transposed ? candidate.index : query.index, column = transposed ? query.index : candidate.index;
            results[row * results_row_stride + column] = (i64)score;
            if ((layout_flags & SZS_LAYOUT_SYMMETRIC) && candidate.index != query.index)
                results[column * results_row_stride + row] = (i64)score;
        }
    }
}

static size_t packed_profile_bytes(u32 classes) { return (size_t)(classes + 1) * (classes + 1) * packed_registers_k * sizeof(u32); }

/** Workgroups that can be RESIDENT at once for this kernel instance with this profile size (never more than the work). */
template <bool local_, bool affine_>
static u32 packed_grid(u64 work_items, u32 classes) {
    static int resident_of[device_slots_k][34]; // per instance, device ordinal and class count
    int *const slot = &resident_of[device_slot()][classes];
    int resident = cached(slot);
    if (!resident) {
        int device = 0, units = 0, per_unit = 0;
        size_t const profile_bytes = packed_profile_bytes(classes);
        if (hipFuncSetAttribute(reinterpret_cast<void const *>(weighted_packed_kernel<local_, affine_>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)packed_profile_bytes(32)) != hipSuccess ||
            hipGetDevice(&device) != hipSuccess ||
            hipDeviceGetAttribute(&units, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess ||
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_unit, weighted_packed_kernel<local_, affine_>,
                                                         (int)packed_block_threads_k, profile_bytes) != hipSuccess ||
            units <= 0 || per_unit <= 0) {
            (void)hipGetLastError();
            units = 256, per_unit = 1;
        }
        resident = units * per_unit;
        remember(slot, resident);
    }
    return (u32)(work_items < (u64)resident ? work_items : (u64)resident);
}

static u64 packed_work_items(u32 queries_count, u32 candidates_count) {
    return (u64)queries_count * ((candidates_count + packed_block_threads_k - 1) / packed_block_threads_k);
}

} // namespace szs_hip

#define SZS_PACKED_DISPATCH(CALL)                                                                                      \
    if (local && affine) CALL(true, true);                                                                             \
    if (local) CALL(true, false);                                                                                      \
    if (affine) CALL(false, true);                                                                                     \
    CALL(false, false);

extern "C" size_t szs_hip_weighted_packed_boundary_bytes(int local, int affine, uint32_t classes, uint32_t queries_count,
                                                         uint32_t candidates_count, uint32_t longest_candidate) {
    using namespace szs_hip;
    if (classes > 32) return 0;
    u64 const work_items = packed_work_items(queries_count, candidates_count);
#define SZS_PACKED_BYTES(...)                                                                                          \
    return packed_header_bytes_k + (size_t)packed_grid<__VA_ARGS__>(work_items, classes) *                             \
                                       (longest_candidate + 1 + packed_boundary_slack_k) * packed_block_threads_k *    \
                                       sizeof(int16_t) * (affine ? 2 : 1)
    SZS_PACKED_DISPATCH(SZS_PACKED_BYTES)
#undef SZS_PACKED_BYTES
}

extern "C" int szs_hip_weighted_packed_scores(int local, int affine, uint32_t classes, szs_cost_model_t const *model,
                                              szs_string_ref_t const *queries, uint32_t queries_count,
                                              szs_string_ref_t const *candidates, uint32_t candidates_count,
                                              uint32_t longest_candidate, int64_t *results, uint64_t results_row_stride,
                                              int layout_flags, void *workspace, void *stream) {
    using namespace szs_hip;
    if (!queries_count || !candidates_count) return 0;
    if (classes > 32) return (int)hipErrorInvalidValue;
    u32 const candidate_blocks = (candidates_count + packed_block_threads_k - 1) / packed_block_threads_k;
    u64 const work_items = packed_work_items(queries_count, candidates_count);
    if (work_items > 0xFFFFFFF0ull) return (int)hipErrorInvalidValue; // the host cuts larger cross-products
    u32 *const counter = static_cast<u32 *>(workspace);
    int16_t *const boundary = reinterpret_cast<int16_t *>(static_cast<char *>(workspace) + packed_header_bytes_k);
    hipStream_t const s = static_cast<hipStream_t>(stream);
    hipError_t const error = hipMemsetAsync(counter, 0, sizeof(u32), s);
    if (error != hipSuccess) return (int)error;
#define SZS_PACKED_LAUNCH(...)                                                                                         \
    {                                                                                                                  \
        u32 const grid = packed_grid<__VA_ARGS__>(work_items, classes);                                                \
        hipLaunchKernelGGL((weighted_packed_kernel<__VA_ARGS__>), dim3(grid), dim3(packed_block_threads_k),            \
                           packed_profile_bytes(classes), s, model, queries, queries_count, candidates, candidates_count, \
                           candidate_blocks, results, results_row_stride, layout_flags, boundary,                      \
                           longest_candidate + 1 + packed_boundary_slack_k, counter, classes);                         \
        return (int)hipGetLastError();                                                                                 \
    }
    SZS_PACKED_DISPATCH(SZS_PACKED_LAUNCH)
#undef SZS_PACKED_LAUNCH
}
