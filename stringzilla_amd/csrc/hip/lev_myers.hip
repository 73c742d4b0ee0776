/*
 *  lev_myers.hip - unit-cost Levenshtein distances (bytes and codepoints) on gfx950, bit-parallel (Myers 1999 / Hyyro 2003).
 *
 *  Replaces, for the ROCm build, the reference's family of unit-cost kernels
 *      unit_myers_*_per_cuda_thread_/warp_   /root/reference/include/stringzillas/similarities/cuda.cuh:2389-2860
 *  and must return exactly what the reference's serial scorer returns
 *      levenshtein_distance_myers<char, serial>   .../similarities/serial.hpp:2073-2314.
 *
 *  MI355X-first design (not a translation of either of the above):
 *
 *  - One workgroup = one QUERY x 256 CANDIDATES; one candidate per lane.  The query is the Myers *pattern*: its
 *    match masks Peq[byte] are built once per workgroup in LDS and shared by all 256 lanes; every lane streams its
 *    own candidate as the *text*.  Work per lane-step is therefore pure register VALU plus one LDS gather.
 *  - 32-bit words, not 64-bit: CDNA4 has no 64-bit integer VALU, so the bit-vector is a W-word big integer whose
 *    `(Eq & VP) + VP` is ONE carry chain (`v_add_co_u32` / `v_addc_co_u32`) and whose shifts are `v_alignbit_b32`.
 *    The reference's CPU form is block-based with explicit HP/HN carries between 64-bit blocks (serial.hpp:2182-2204);
 *    the full-width form computes the same column of the DP matrix, hence the same distance (tests pin this).
 *  - The pattern is RIGHT-aligned in its W words: position i sits at bit (32 W - len + i).  The `pad` low bits are
 *    phantom rows initialised VP = VN = Eq = 0; they stay inert (HP = 1, HN = 0, VP = VN = 0 forever) and hand the
 *    `+1 per column` top-row boundary into the first real bit through the ordinary shift.  One kernel body serves every
 *    query whose length fits in W words (a zero-length query degenerates to `distance = candidate length` by itself).
 *  - No per-column score tracking: when a lane has consumed its text, distance = len(text) + popcount(VP) - popcount(VN)
 *    (row deltas telescope; phantom rows contribute nothing).  Lanes that finish before their wavefront's longest text
 *    are frozen by the EXEC mask.  Candidates are length-sorted by the host, so the 64 lanes of a wave finish within
 *    a few columns of each other and the predicated part of the loop is short.
 *  - Results are written straight to results[query.index * stride + candidate.index] (+ mirror when symmetric).
 *
 *  - Queries of up to 256 bytes - whatever mix of lengths - are ONE launch: the word count is a per-workgroup scalar
 *    decision inside `levenshtein_myers_short_kernel`, so a batch straddling several widths pays neither a launch per
 *    width nor padding to the widest.  Longer queries use one instantiated width per launch (up to 64 words), and
 *    beyond 2048 bytes horizontal strips of equal width whose last-row deltas are parked per text column (further down).
 *  - Codepoint twins of all of it up to 2048 runes: UTF-32 texts, Peq keyed through a rune table in LDS (slots for the
 *    short kernel, dense ids for the long ones) - the reference's open-addressing Peq (serial.hpp:2328-2512) on a GPU.
 *
 *  Cost per text byte per lane (ISA count, W = 4): 48 VALU = 10.5 per word + 2 history pushes + LDS addressing, i.e.
 *  0.38 VALU per DP cell; LDS: one ds_read_b128 per four words.  HBM: tapes once + 8 B per result; see DESIGN.md
 *  section 5 for the roofline arithmetic and scripts/valu_peak.hip for the measured VALU ceiling of this mix.
 */
#include "myers_core.hpp"

#include <cstdlib>

namespace szs_hip {

#ifndef SZS_MYERS_SHORT_TEXT_DWORDS
#define SZS_MYERS_SHORT_TEXT_DWORDS 2 // text dwords per main-loop iteration of the short-query bodies: 8 columns (16 left ~7 columns per lane to the predicated tail; measured +1.3 % on config 2)
#endif

// (match-mask layout, the column update and the strip column: hip/myers_core.hpp - shared with hip/myers_queue.hip)

/**
 *  One workgroup: one query against 256 candidates, one candidate per lane.
 *
 *  The distance is never accumulated column by column.  Row deltas telescope: D[m][j] = D[0][j] + sum_i (VP_i - VN_i),
 *  so once a lane has consumed its whole text, distance = text length + popcount(VP) - popcount(VN) (phantom rows hold
 *  VP = VN = 0 and drop out).  A lane whose text has ended must therefore FREEZE its state while longer texts of the
 *  same wavefront go on - which is what the EXEC mask does for free:
 *    - main loop: all 64 lanes still have a full batch of columns left -> no predication, no per-column bookkeeping;
 *    - ragged tail (from the wavefront's shortest text to its longest): every column is predicated on
 *      `column < text length`, finished lanes are simply masked off.
 *  Candidates arrive length-sorted, so the ragged tail is a handful of columns unless the batch itself is ragged.
 *
 *  @tparam words_        32-bit words of the pattern bit-vector; the query fits in 32 * words_ symbols.
 *  @tparam text_dwords_  text dwords consumed per main-loop iteration: 4 for short patterns, 1 for long ones to keep
 *                        the unrolled body inside the instruction cache.  Bytes: 4 columns per dword; runes: 1.
 *  @tparam runes_        symbols are UTF-32 codepoints (strings are `u32` arrays, lengths count runes) and Peq is keyed
 *                        through the workgroup's rune table `keys` instead of directly by byte value.
 */
template <int words_, int text_dwords_, bool runes_>
__device__ __forceinline__ void myers_workgroup(u32 *peq, u32 *keys, szs_string_ref_t const query,
                                                szs_string_ref_t const *__restrict__ candidates, u32 candidates_count,
                                                u32 candidate_block, u64 *__restrict__ results, u64 results_row_stride,
                                                int symmetric, szs_ref_guard_t const &guard, u32 alphabet = 0,
                                                u32 *claimed_rows = nullptr, u32 blocks_here = 1) {
    constexpr int rows = runes_ ? rune_slots_k : byte_rows_k;
    using layout = peq_layout<words_, rows>;
    if (guard.enabled && !ref_is_current(guard, 0, query)) { // refs of an earlier call: this query's no longer holds - uniform
        if (threadIdx.x == 0) *guard.stale = guard.sequence;
        return;
    }
    u32 const query_length = query.length;
    u32 const pad = 32u * words_ - query_length; // phantom low rows

    // ---- Peq: zero, then scatter the pattern's bits (LDS atomics; a 128-symbol query is 128 ORs per workgroup).
    for (int i = threadIdx.x; i < layout::total_dwords; i += 256) peq[i] = 0;
    if constexpr (runes_) {
        if (alphabet) { // a renumbered batch (utf8.hip): `keys` is a direct table, id -> Peq row (0: not in the pattern)
            for (u32 i = threadIdx.x; i <= alphabet; i += 256) keys[i] = 0;
            if (threadIdx.x == 0) *claimed_rows = 0;
        }
        else
            for (int i = threadIdx.x; i < rune_slots_k; i += 256) keys[i] = rune_slot_empty_k;
    }
    __syncthreads();
    if (runes_ && alphabet) {
        u32 const *pattern = reinterpret_cast<u32 const *>(query.address);
        for (u32 i = threadIdx.x; i < query_length; i += 256) { // the first thread to meet an id takes the next row for it
            u32 const id = pattern[i];
            if (atomicCAS(&keys[id], 0u, ~0u) == 0u) keys[id] = atomicAdd(claimed_rows, 1u) + 1;
        }
        __syncthreads();
        for (u32 i = threadIdx.x; i < query_length; i += 256) {
            u32 const position = pad + i;
            atomicOr(&peq[layout::dword_index((int)keys[pattern[i]], (int)(position >> 5))], 1u << (position & 31));
        }
    }
    else if constexpr (runes_) {
        u32 const *pattern = reinterpret_cast<u32 const *>(query.address);
        for (u32 i = threadIdx.x; i < query_length; i += 256) {
            u32 const rune = pattern[i];
            u32 slot = rune_slot_hash(rune); // claim the rune's slot: at most 256 distinct runes in 512 slots
            for (;;) {
                u32 const previous = atomicCAS(&keys[slot], rune_slot_empty_k, rune);
                if (previous == rune_slot_empty_k || previous == rune) break;
                slot = (slot + 1) & (rune_slots_k - 1);
            }
            u32 const position = pad + i;
            atomicOr(&peq[layout::dword_index((int)slot, (int)(position >> 5))], 1u << (position & 31));
        }
    }
    else {
        u8 const *pattern = reinterpret_cast<u8 const *>(query.address);
        for (u32 i = threadIdx.x; i < query_length; i += 256) {
            u32 const position = pad + i;
            atomicOr(&peq[layout::dword_index(pattern[i], (int)(position >> 5))], 1u << (position & 31));
        }
    }
    __syncthreads();

    // ---- `blocks_here` candidate blocks against this table, the heaviest (`candidate_block`) first: when a launch is tens of
    //      thousands of workgroups of tiny strings, the table and the round trips to the query are shared by several blocks
    //      (the launcher decides: launch_myers).
#pragma unroll 1
    for (u32 block_step = 0; block_step < blocks_here; ++block_step) {
    // ---- this lane's candidate
    u32 const candidate_slot = (candidate_block - block_step) * SZS_CANDIDATES_PER_WORKGROUP + threadIdx.x;
    bool live = candidate_slot < candidates_count;
    szs_string_ref_t candidate = {0, 0, 0};
    if (live) candidate = candidates[candidate_slot];
    if (live && guard.enabled && !ref_is_current(guard, 1, candidate)) { // a stale candidate ref is never dereferenced
        *guard.stale = guard.sequence;
        live = false;
    }
    if ((symmetric & SZS_LAYOUT_SYMMETRIC) && candidate.index > query.index) live = false; // upper triangle: mirrored from below
    u32 const text_length = live ? candidate.length : 0;
    u32 const longest_in_wave = wave_max_u32(text_length);
    u32 const shortest_in_wave = ~wave_max_u32(live ? ~text_length : 0u); // over live lanes; no live lane: ~0, unused

    u32 vp[words_], vn[words_];
#pragma unroll
    for (int w = 0; w < words_; ++w) {
        u32 const first_bit = 32u * w;
        vp[w] = first_bit >= pad ? ~0u : (first_bit + 32u <= pad ? 0u : (~0u << (pad - first_bit)));
        vn[w] = 0;
    }

    // One DP column: the match masks of `symbol` (a byte, or a rune looked up through the rune table) update VP / VN.
    auto take = [&](u32 symbol) {
        u32 eq[words_];
        load_match_masks<words_, rows>(peq, !runes_ ? symbol : alphabet ? keys[symbol] : find_rune_slot(keys, symbol), eq);
        myers_column<words_>(vp, vn, eq);
    };

    u32 column = 0;
    if constexpr (runes_) {
        // ---- codepoints: FOUR columns per 16-byte load.  Every lane streams its own string, so a load instruction touches 64
        //      cache lines whatever its width; a dword per column made the rune kernels wait on the memory pipeline (config 5u:
        //      135 GB of line traffic per call, VALU issue 56 % busy).  The transcoder starts every string on a 16-byte boundary.
        u32 const *const runes = reinterpret_cast<u32 const *>(candidate.address);
        auto quad_at = [&](u32 index) -> uint4 { // a multiple of 4; symbols past the text's end are never consumed
            return index < text_length ? *reinterpret_cast<uint4 const *>(runes + index) : make_uint4(0, 0, 0, 0);
        };
        if (4 <= shortest_in_wave && longest_in_wave) {
            uint4 ahead = quad_at(0);
            for (; column + 4 <= shortest_in_wave; column += 4) {
                uint4 const now = ahead;
                ahead = quad_at(column + 4);
                take(now.x), take(now.y), take(now.z), take(now.w);
            }
        }
#pragma unroll 1
        for (; column < longest_in_wave; ++column)
            if (column < text_length) take(runes[column]);
    }
    else {
        // ---- bytes: `raw_low` is text dword `dword`; `ahead[d]` is text dword `dword + 1 + d`, loaded one iteration early.
        // Lanes without a text stream from the query instead (always-valid memory; their symbols are never consumed), so
        // the main loop's reads are index-clamped, not predicated: no branch splits the unrolled batch.
        u64 const safe_address = text_length ? candidate.address : query.address;
        text_stream_t text(safe_address, text_length);
        if (!text_length) text.valid_dwords = query_length ? 1 : 0;
        u32 raw_low = text.raw(0);
        u32 dword = 0;
        constexpr u32 columns_per_iteration = 4 * text_dwords_;
#ifndef SZS_MYERS_CLAMPED_READS
#define SZS_MYERS_CLAMPED_READS(TEXT_DWORDS) ((TEXT_DWORDS) == 1)
#endif
        // Measured on MI355X: index-clamped (branch-free) reads help the one-dword loops of the long kernels; the
        // 16-column batches of the short kernels schedule better with the predicated reads.
        constexpr bool clamped_reads = SZS_MYERS_CLAMPED_READS(text_dwords_);
        if (columns_per_iteration <= shortest_in_wave && longest_in_wave && query_length) {
            u32 ahead[text_dwords_];
#pragma unroll
            for (int d = 0; d < text_dwords_; ++d) ahead[d] = clamped_reads ? text.raw_clamped(1 + d) : text.raw(1 + d);
            for (; column + columns_per_iteration <= shortest_in_wave;
                 column += columns_per_iteration, dword += text_dwords_) {
                u32 symbols[text_dwords_];
                symbols[0] = text.splice(raw_low, ahead[0]);
#pragma unroll
                for (int d = 1; d < text_dwords_; ++d) symbols[d] = text.splice(ahead[d - 1], ahead[d]);
                raw_low = ahead[text_dwords_ - 1];
                // Issue the next iteration's loads now; they retire under the VALU work below.
#pragma unroll
                for (int d = 0; d < text_dwords_; ++d)
                    ahead[d] = clamped_reads ? text.raw_clamped(dword + text_dwords_ + 1 + d)
                                             : text.raw(dword + text_dwords_ + 1 + d);
#pragma unroll
                for (int step = 0; step < 4 * text_dwords_; ++step) take((symbols[step / 4] >> (8 * (step % 4))) & 0xFFu);
            }
        }
#ifndef SZS_MYERS_MID_LOOP
#define SZS_MYERS_MID_LOOP 0
#endif
        // ---- between the two (build variant): whole dwords that every live lane still has, one at a time, unpredicated -
        // the batch above leaves up to 4 * text_dwords_ - 1 columns to the per-column predicates of the tail otherwise.
        if constexpr (SZS_MYERS_MID_LOOP && text_dwords_ > 1) {
            if (column + 4 <= shortest_in_wave && longest_in_wave && query_length) {
                u32 ahead_one = text.raw(dword + 1);
#pragma unroll 1
                for (; column + 4 <= shortest_in_wave; column += 4, ++dword) {
                    u32 const symbols = text.splice(raw_low, ahead_one);
                    raw_low = ahead_one;
                    ahead_one = text.raw(dword + 2);
#pragma unroll
                    for (int step = 0; step < 4; ++step) take((symbols >> (8 * step)) & 0xFFu);
                }
            }
        }
        // ---- ragged tail: one dword (4 columns) per iteration, each column predicated on this lane's own length.
        if (column < longest_in_wave) {
            u32 next = text.raw(dword + 1);
#pragma unroll 1
            for (; column < longest_in_wave; column += 4, ++dword) {
                u32 const after = text.raw(dword + 2);
                u32 const symbols = text.splice(raw_low, next);
                raw_low = next, next = after;
#pragma unroll
                for (int step = 0; step < 4; ++step)
                    if (column + step < text_length) take((symbols >> (8 * step)) & 0xFFu);
            }
        }
    }

    if (live) {
        u32 distance = text_length;
#pragma unroll
        for (int w = 0; w < words_; ++w) distance += (u32)__builtin_popcount(vp[w]) - (u32)__builtin_popcount(vn[w]);
        bool const transposed = (symmetric & SZS_LAYOUT_TRANSPOSED) != 0; // kernel roles swapped by the host
        u64 const row = transposed ? candidate.index : query.index, column = transposed ? query.index : candidate.index;
        results[row * results_row_stride + column] = distance;
        if ((symmetric & SZS_LAYOUT_SYMMETRIC) && candidate.index != query.index)
            results[column * results_row_stride + row] = distance;
    }
    } // the next candidate block
}

/** Workgroup -> (query, candidate block).  Candidates arrive ascending, so walking the candidate blocks backwards hands out
 *  the heaviest work first and the launch ends on its lightest workgroups.  BLOCK-major: the workgroups of one candidate
 *  block - every query against the same 256 texts - are neighbours, and a CU's resident workgroups are queries of different
 *  widths that finish at different times, so one's Peq build overlaps another's columns.  Query-major (all blocks of a query
 *  side by side, a CU's residents in lock step) was 15 % slower on config 2 (200 -> 171 us), 19 % on config 5 (11.5 -> 9.6 ms). */
#ifndef SZS_WORK_BLOCK_MAJOR
#define SZS_WORK_BLOCK_MAJOR 1
#endif
__device__ __forceinline__ void myers_work_item(u32 candidate_blocks, u32 &query_slot, u32 &candidate_block) {
#if SZS_WORK_BLOCK_MAJOR
    u32 const queries_count = gridDim.x / candidate_blocks;
    query_slot = blockIdx.x % queries_count;
    candidate_block = candidate_blocks - 1 - blockIdx.x / queries_count;
#else
    query_slot = blockIdx.x / candidate_blocks;
    candidate_block = candidate_blocks - 1 - blockIdx.x % candidate_blocks;
#endif
}

/** Long queries (more than 8 words): every query of the launch uses the same instantiated width. */
#ifndef SZS_MYERS_LONG_WAVES
#define SZS_MYERS_LONG_WAVES 2 // two wavefronts per SIMD at every width: 64 words then cost 11 spilled registers and still
#endif                         // run 1.3x faster than one wavefront with none (profiles/r01/myers_long_occupancy_v1.txt)
template <int words_>
__global__ __launch_bounds__(256, SZS_MYERS_LONG_WAVES) void levenshtein_myers_long_kernel(szs_string_ref_t const *__restrict__ queries,
                                                                      szs_string_ref_t const *__restrict__ candidates,
                                                                      u32 candidates_count, u32 candidate_blocks,
                                                                      u64 *__restrict__ results, u64 results_row_stride,
                                                                      int symmetric, szs_ref_guard_t guard) {
    __shared__ __attribute__((aligned(16))) u32 peq[peq_layout<words_>::total_dwords];
    u32 query_slot, candidate_block;
    myers_work_item(candidate_blocks, query_slot, candidate_block);
    myers_workgroup<words_, 1, false>(peq, nullptr, queries[query_slot], candidates, candidates_count, candidate_block,
                                      results, results_row_stride, symmetric, guard);
}

/* ---- the planner folded into the short launch (kernels.h: szs_fused_plan_t) --------------------------------------------
 *
 *  What hip/planner.hip does in a launch of its own - tape offsets in, length-sorted refs out - done by ONE workgroup of the
 *  scoring launch per side: 1024 length bins in the LDS that becomes the match masks afterwards, 1024 staged refs beside it.
 *  The sort is a counting sort; strings of equal length land in whatever order the LDS atomics decide, which is why ONE
 *  workgroup sorts a side for everybody (two sorters of the same side could disagree on the blocks of 256).
 */
__device__ __forceinline__ u64 fused_offset(void const *offsets, u32 wide, u64 index) {
    return wide ? static_cast<u64 const *>(offsets)[index] : (u64) static_cast<u32 const *>(offsets)[index];
}

__device__ __forceinline__ void fused_sort_side(szs_plan_side_t const &side, u32 is_query_side, u32 sequence, u32 *ready, u32 second_ready,
                                                u32 withhold, szs_fused_side_report_t *report, u32 *histogram /* SZS_FUSED_BINS dwords of LDS */,
                                                szs_string_ref_t *staged /* SZS_FUSED_MOST_STRINGS refs of LDS */) {
    constexpr u32 per_thread = SZS_FUSED_MOST_STRINGS / 256;
    u32 const tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6, count = side.count;
    bool const one_pass = count <= SZS_FUSED_MOST_STRINGS; // every string of the side in this workgroup's registers at once
    __shared__ u32 wave_bins[4], wave_longest[4], wave_status[4];
    __shared__ unsigned long long wave_symbols[4], wave_bands_systolic[4], wave_bands_chain[4], wave_squares[4];

    u64 const began = wall_clock64();
    for (u32 bin = tid; bin < SZS_FUSED_BINS; bin += 256) histogram[bin] = 0;
    u64 from[per_thread], to[per_thread]; // one pass: all loads of the side in flight at once - ONE round trip to memory for the whole sort
#pragma unroll
    for (u32 k = 0; k < per_thread; ++k) {
        u32 const i = tid + k * 256;
        from[k] = to[k] = 0;
        if (one_pass && i < count) from[k] = fused_offset(side.offsets, side.wide, i), to[k] = fused_offset(side.offsets, side.wide, (u64)i + 1);
    }
    __syncthreads();
    u32 bins[per_thread];
    u32 status = 0, longest = 0;
    u64 symbols = 0, bands_systolic = 0, bands_chain = 0, squares = 0;
    // what one string adds to the side's figures; its bin, or ~0 when its offsets are malformed
    auto count_in = [&](u64 first, u64 last) -> u32 {
        if (last < first) return status |= SZS_PLAN_STATUS_DESCENDING, ~0u;
        if (last - first > 0xFFFFFFFFull) return status |= SZS_PLAN_STATUS_OVERFLOW, ~0u;
        u32 const length = (u32)(last - first);
        symbols += length, squares += (u64)length * length, longest = length > longest ? length : longest;
        bands_systolic += length ? (length + SZS_SYSTOLIC_BAND_ROWS - 1) / SZS_SYSTOLIC_BAND_ROWS : 1;
        bands_chain += length ? (length + SZS_MYERS_CHAIN_BAND_ROWS - 1) / SZS_MYERS_CHAIN_BAND_ROWS : 1;
        u32 const bin = length < SZS_FUSED_BINS - 1 ? length : SZS_FUSED_BINS - 1;
        atomicAdd(&histogram[bin], 1u);
        return bin;
    };
    if (one_pass) {
#pragma unroll
        for (u32 k = 0; k < per_thread; ++k) {
            bins[k] = ~0u;
            if (tid + k * 256 < count) bins[k] = count_in(from[k], to[k]);
        }
    }
    else { // a larger side (round 6): counted in one walk over its offsets, 256 strings at a time, placed in a second one below
#pragma unroll 1
        for (u32 first = 0; first < count; first += 1024) {
            u64 some_from[4], some_to[4];
#pragma unroll
            for (u32 k = 0; k < 4; ++k) {
                u32 const i = first + tid + k * 256;
                some_from[k] = some_to[k] = 0;
                if (i < count) some_from[k] = fused_offset(side.offsets, side.wide, i), some_to[k] = fused_offset(side.offsets, side.wide, (u64)i + 1);
            }
#pragma unroll
            for (u32 k = 0; k < 4; ++k)
                if (first + tid + k * 256 < count) (void)count_in(some_from[k], some_to[k]);
        }
    }
    if (tid == 0) report->ticks[0] = (u32)began, report->ticks[1] = (u32)(wall_clock64() - began);
    // (only what the refs depend on is reduced before they are published - malformed offsets, the longest string; the sums the
    // host wants follow behind the publication: everybody else is waiting for the refs, nobody for the report)
#pragma unroll
    for (int offset = 32; offset >= 1; offset >>= 1) {
        status |= (u32)__shfl_xor((int)status, offset, 64);
        u32 const other = (u32)__shfl_xor((int)longest, offset, 64);
        longest = other > longest ? other : longest;
    }
    if (lane == 0) wave_status[wave] = status, wave_longest[wave] = longest;
    __syncthreads();
    // ---- bins become positions: thread t owns bins [4 t, 4 t + 4)
    u32 mine[4], sum = 0;
#pragma unroll
    for (u32 k = 0; k < 4; ++k) mine[k] = histogram[4 * tid + k], sum += mine[k];
    u32 inclusive = sum;
#pragma unroll
    for (int offset = 1; offset < 64; offset <<= 1) {
        u32 const other = (u32)__shfl_up((int)inclusive, offset, 64);
        if (lane >= (u32)offset) inclusive += other;
    }
    if (lane == 63) wave_bins[wave] = inclusive;
    __syncthreads();
    status = wave_status[0] | wave_status[1] | wave_status[2] | wave_status[3];
    longest = wave_longest[0];
#pragma unroll
    for (u32 w = 1; w < 4; ++w) longest = wave_longest[w] > longest ? wave_longest[w] : longest;
    u32 running = inclusive - sum;
    for (u32 w = 0; w < wave; ++w) running += wave_bins[w];
#pragma unroll
    for (u32 k = 0; k < 4; ++k) histogram[4 * tid + k] = running, running += mine[k];
    __syncthreads();
    if (tid == 0) report->ticks[2] = (u32)(wall_clock64() - began);
    // ---- every string's ref lands at its position (a malformed side sorts nothing: blank refs in the caller's order);
    //      strings of one length take their places in whatever order the atomics decide
    u32 const blank = status || (is_query_side && longest > 32u * SZS_MYERS_SHORT_WORDS);
    if (one_pass) { // ... in LDS first, and leaves it in order: position p and count - 1 - p, whole lines per wavefront
#pragma unroll
        for (u32 k = 0; k < per_thread; ++k) {
            u32 const i = tid + k * 256;
            if (i >= count) continue;
            szs_string_ref_t ref;
            ref.address = status ? side.base : side.base + from[k], ref.length = blank ? 0u : (u32)(to[k] - from[k]), ref.index = i;
            staged[status ? i : atomicAdd(&histogram[bins[k]], 1u)] = ref;
        }
        __syncthreads();
        for (u32 position = tid; position < count; position += 256) {
            szs_string_ref_t const ref = staged[position];
            side.ascending[position] = ref, side.descending[count - 1 - position] = ref;
        }
    }
    else { // ... straight in memory: the second walk over the offsets (they come from the L2 now), sixteen bytes a ref, scattered
#pragma unroll 1
        for (u32 first = 0; first < count; first += 1024) {
            u64 some_from[4], some_to[4];
#pragma unroll
            for (u32 k = 0; k < 4; ++k) {
                u32 const i = first + tid + k * 256;
                some_from[k] = some_to[k] = 0;
                if (i < count) some_from[k] = fused_offset(side.offsets, side.wide, i), some_to[k] = fused_offset(side.offsets, side.wide, (u64)i + 1);
            }
#pragma unroll
            for (u32 k = 0; k < 4; ++k) {
                u32 const i = first + tid + k * 256;
                if (i >= count) continue;
                u32 const length = status ? 0u : (u32)(some_to[k] - some_from[k]);
                szs_string_ref_t ref;
                ref.address = status ? side.base : side.base + some_from[k], ref.length = blank ? 0u : length, ref.index = i;
                u32 const position = status ? i : atomicAdd(&histogram[length < SZS_FUSED_BINS - 1 ? length : SZS_FUSED_BINS - 1], 1u);
                side.ascending[position] = ref, side.descending[count - 1 - position] = ref;
            }
        }
    }
    // Every thread's stores have reached the L2 behind the barrier (it waits for them; the L1 writes through); ONE agent-scope
    // release by one thread then writes the L2 back before the word that says so changes.
    __syncthreads();
    if (tid == 0) {
        report->ticks[3] = (u32)(wall_clock64() - began);
        if (!withhold) {
            __hip_atomic_store(ready, sequence, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            if (second_ready) __hip_atomic_store(ready + 32, sequence, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // a symmetric call: both roles
        }
        report->ticks[4] = (u32)(wall_clock64() - began);
    }
    // ---- the report (the host reads it when the launch has ended)
#pragma unroll
    for (int offset = 32; offset >= 1; offset >>= 1) {
        symbols += ((u64)(u32)__shfl_xor((int)(u32)(symbols >> 32), offset, 64) << 32) | (u32)__shfl_xor((int)(u32)symbols, offset, 64);
        squares += ((u64)(u32)__shfl_xor((int)(u32)(squares >> 32), offset, 64) << 32) | (u32)__shfl_xor((int)(u32)squares, offset, 64);
        bands_systolic += ((u64)(u32)__shfl_xor((int)(u32)(bands_systolic >> 32), offset, 64) << 32) | (u32)__shfl_xor((int)(u32)bands_systolic, offset, 64);
        bands_chain += ((u64)(u32)__shfl_xor((int)(u32)(bands_chain >> 32), offset, 64) << 32) | (u32)__shfl_xor((int)(u32)bands_chain, offset, 64);
    }
    if (lane == 0) wave_symbols[wave] = symbols, wave_bands_systolic[wave] = bands_systolic, wave_bands_chain[wave] = bands_chain, wave_squares[wave] = squares;
    if (tid <= SZS_PLAN_RANK_SAMPLES) { // the length at 33 ranks of the side (what the queue order is planned from)
        u32 const rank = (u32)((u64)tid * (count ? count - 1 : 0) / SZS_PLAN_RANK_SAMPLES);
        // (the larger sides read their own stores back: the same CU wrote them, behind the barrier above)
        report->rank_lengths[tid] = status || !count ? 0u : one_pass ? staged[rank].length : side.ascending[rank].length;
    }
    __syncthreads();
    if (tid == 0) {
        report->status = status, report->blank = blank, report->reserved = 0;
        report->stats.count = count, report->stats.longest = longest;
        report->stats.symbols = wave_symbols[0] + wave_symbols[1] + wave_symbols[2] + wave_symbols[3];
        report->stats.bands_systolic = wave_bands_systolic[0] + wave_bands_systolic[1] + wave_bands_systolic[2] + wave_bands_systolic[3];
        report->stats.bands_chain = wave_bands_chain[0] + wave_bands_chain[1] + wave_bands_chain[2] + wave_bands_chain[3];
        report->squares = wave_squares[0] + wave_squares[1] + wave_squares[2] + wave_squares[3];
        report->sequence = sequence;
    }
}

/** Sorters sort, everybody waits for both sides.  Workgroups are dispatched in order: 0 and 1 never wait for anyone.
 *  False: this workgroup ran out of polls (the sorters are not resident, or one of them never published) - it must not touch a ref. */
__device__ __forceinline__ bool fused_prologue(szs_fused_plan_t const &plan, u32 *scratch) {
    // 16 KB of LDS that only the two sorting workgroups touch: the scoring bodies keep five workgroups per CU either way (95 VGPRs)
    __shared__ __attribute__((aligned(16))) szs_string_ref_t staged[SZS_FUSED_MOST_STRINGS];
    for (u32 s = 0; s < (plan.symmetric ? 1u : 2u); ++s) // a symmetric call: one side, sorted once, serves both roles
        if (blockIdx.x == s % gridDim.x) {
            fused_sort_side(plan.side[s], s == 0 || plan.symmetric, plan.sequence, plan.ready + 32 * s, plan.symmetric, plan.withhold, plan.report + s,
                            scratch, staged);
            __syncthreads(); // the LDS is sorted in again (a grid of one workgroup), then becomes the match masks
        }
    // The wait is a RELAXED load at agent scope (it goes to the device's coherence point every time) and the barrier orders the
    // workgroup behind it.  No acquire FENCE: at agent scope that is a `buffer_inv sc1` - it drops the L2 of the XCD - and 4096
    // workgroups doing that as they start made every text read miss (0.41 ms where the plain launch takes 0.17).  None is needed:
    // nobody reads a ref before its side is published, so no cache of this launch can hold a stale one (the caches start a
    // launch empty), and the sorter's release wrote its lines back before the word changed.
    // (Sleeping through most of the sort before the first poll changed nothing: the polls are not what the launch waits for.)
    // Measured and not kept: the waiting workgroups pulling both tapes into their XCD's L2 meanwhile (a line per thread) - the
    // sorters' own first loads then took 2.3 us instead of 1.4 and the launch 181.6 us instead of 179.6.
    // (What the wait costs, measured with build variants that skip it - reading the previous call's refs, results garbage: the
    // launch 179.6 us; without the wait 173.0; without the sort as well 171.5, the plain launch's time.)
    // The wait is BOUNDED (round 6): `poll_budget` polls of ~1 us each, four orders of magnitude beyond what a sorter takes.  A
    // workgroup whose polls run out says so in pinned memory and scores nothing - it leaves the CU to whoever is queued behind it
    // (the sorters, should the dispatch order ever not be the observed one) - and the host plans the call the ordinary way.
    __shared__ u32 unpublished[2];
    if (threadIdx.x < 2) {
        u32 polls = 0;
        bool published;
        while (!(published = __hip_atomic_load(plan.ready + 32 * threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == plan.sequence) &&
               ++polls < plan.poll_budget)
            __builtin_amdgcn_s_sleep(8);
        unpublished[threadIdx.x] = !published;
    }
    __syncthreads();
    if (unpublished[0] | unpublished[1]) {
        if (threadIdx.x == 0) __hip_atomic_store(plan.gave_up, plan.sequence, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        return false;
    }
    return true;
}

#ifndef SZS_MYERS_SHORT_WAVES
#define SZS_MYERS_SHORT_WAVES 1
#endif
/**
 *  Short queries (up to 256 symbols = 8 words), ANY mix of lengths in ONE launch: the width is a per-workgroup (scalar)
 *  decision, so every query runs at exactly ceil(length / 32) words, and a batch whose queries straddle several widths
 *  needs neither one launch per width nor padding to a common width.  All eight byte bodies fit the 64-VGPR budget.
 *  `runes_`: the codepoint-level twin - strings are UTF-32 arrays produced by utf8.hip, Peq is keyed by a rune table.
 */
template <bool runes_, bool merged_, bool fused_ = false>
__device__ __forceinline__ void myers_short_body(
    szs_string_ref_t const *__restrict__ queries, szs_string_ref_t const *__restrict__ candidates, u32 candidates_count,
    u32 candidate_blocks, u64 *__restrict__ results, u64 results_row_stride, int symmetric, szs_ref_guard_t guard, u32 alphabet,
    u32 blocks_per_group, szs_fused_plan_t const *fused = nullptr) {
    __shared__ __attribute__((aligned(16))) u32 peq[peq_layout<8, runes_ ? rune_slots_k : byte_rows_k>::total_dwords];
    if constexpr (fused_) {
        static_assert(peq_layout<8, byte_rows_k>::total_dwords >= SZS_FUSED_BINS, "the sort's histogram borrows the masks' LDS");
        if (!fused_prologue(*fused, peq)) return; // the refs `queries` / `candidates` point at exist from here on
    }
    __shared__ u32 slot_keys[runes_ ? rune_slots_k : 1];
    __shared__ u32 claimed_rows;
    extern __shared__ u32 rows_of_ids[]; // runes of a renumbered batch: alphabet + 1 dwords of dynamic LDS
    u32 *const keys = runes_ && alphabet ? rows_of_ids : slot_keys;
    // `candidate_blocks` counts GROUPS of `blocks_per_group` blocks here (1 unless the launcher merged them); group g holds the
    // blocks [g x n, (g + 1) x n) that exist, and the workgroup walks them downwards from the last
    u32 query_slot, candidate_group;
    myers_work_item(candidate_blocks, query_slot, candidate_group);
    u32 const all_blocks = (candidates_count + SZS_CANDIDATES_PER_WORKGROUP - 1) / SZS_CANDIDATES_PER_WORKGROUP;
    u32 const first_block = merged_ ? candidate_group * blocks_per_group : candidate_group;
    // (a compile-time 1 without merging: the block loop of myers_workgroup folds away - it cost config 2 1.5 % as a real loop)
    u32 const blocks_here = !merged_ ? 1u : all_blocks - first_block < blocks_per_group ? all_blocks - first_block : blocks_per_group;
    u32 const candidate_block = first_block + blocks_here - 1;
    szs_string_ref_t query = queries[query_slot];
    if constexpr (fused_) {
        // The refs are written during this very launch, so the compiler loads this one with a VECTOR load (no `s_load` from memory
        // the kernel may clobber) and the four dwords - uniform over the workgroup - sat in VGPRs through the whole body: under the
        // 96-register bound eight registers a lane were spilled, and those scratch stores were the "4x write amplification" of round 5's
        // counters (33.4 MB written per launch for 8.4 MB of results: 4096 workgroups x 256 lanes x 24 bytes of spills).
        query.address = ((u64)(u32)__builtin_amdgcn_readfirstlane((int)(u32)(query.address >> 32)) << 32) |
                        (u32)__builtin_amdgcn_readfirstlane((int)(u32)query.address);
        query.length = (u32)__builtin_amdgcn_readfirstlane((int)query.length);
        query.index = (u32)__builtin_amdgcn_readfirstlane((int)query.index);
    }
    u32 const words = __builtin_amdgcn_readfirstlane(query.length ? (query.length + 31u) / 32u : 1u);
#define SZS_MYERS_BODY(W)                                                                                              \
    case W:                                                                                                            \
        myers_workgroup<W, SZS_MYERS_SHORT_TEXT_DWORDS, runes_>(peq, keys, query, candidates, candidates_count,       \
                                                                candidate_block, results, results_row_stride,         \
                                                                symmetric, guard, alphabet, &claimed_rows, blocks_here); \
        break;
    switch (words) {
        SZS_MYERS_BODY(1)
        SZS_MYERS_BODY(2)
        SZS_MYERS_BODY(3)
        SZS_MYERS_BODY(4)
        SZS_MYERS_BODY(5)
        SZS_MYERS_BODY(6)
        SZS_MYERS_BODY(7)
    default: // 8; the host never sends longer queries here
        myers_workgroup<8, SZS_MYERS_SHORT_TEXT_DWORDS, runes_>(peq, keys, query, candidates, candidates_count,
                                                                candidate_block, results, results_row_stride, symmetric, guard,
                                                                alphabet, &claimed_rows, blocks_here);
        break;
    }
#undef SZS_MYERS_BODY
}

template <bool runes_>
__global__ __launch_bounds__(256, SZS_MYERS_SHORT_WAVES) void levenshtein_myers_short_kernel(
    szs_string_ref_t const *__restrict__ queries, szs_string_ref_t const *__restrict__ candidates, u32 candidates_count,
    u32 candidate_blocks, u64 *__restrict__ results, u64 results_row_stride, int symmetric, szs_ref_guard_t guard, u32 alphabet,
    u32 blocks_per_group) {
    myers_short_body<runes_, false>(queries, candidates, candidates_count, candidate_blocks, results, results_row_stride, symmetric, guard,
                                    alphabet, blocks_per_group);
}

/** The planner folded in (kernels.h: szs_fused_plan_t): the refs are written by workgroups 0 and 1 of this very launch. */
#ifndef SZS_MYERS_FUSED_WAVES
#define SZS_MYERS_FUSED_WAVES 5 // the plain short kernel's five wavefronts per SIMD (95 VGPRs); unbounded, the prologue makes it 98
#endif
template <bool merged_>
__global__ __launch_bounds__(256, SZS_MYERS_FUSED_WAVES) void levenshtein_myers_short_fused_kernel(szs_fused_plan_t plan, u32 candidate_blocks,
                                                                                                   u64 *results, u64 results_row_stride,
                                                                                                   int layout) {
    szs_ref_guard_t const none = {};
    // (plain pointers, not __restrict__ const ones: these arrays ARE written during the launch, by the sorting workgroups)
    // a symmetric call (round 6): the one side in both roles - its descending refs the patterns, its ascending refs the texts
    szs_plan_side_t const &texts = plan.symmetric ? plan.side[0] : plan.side[1];
    myers_short_body<false, merged_, true>(plan.side[0].descending, texts.ascending, texts.count, candidate_blocks, results, results_row_stride,
                                           layout, none, 0u, plan.blocks_per_group, &plan);
}

/** The same with `blocks_per_group` candidate blocks per workgroup (launch_myers_short decides). */
template <bool runes_>
__global__ __launch_bounds__(256, SZS_MYERS_SHORT_WAVES) void levenshtein_myers_short_merged_kernel(
    szs_string_ref_t const *__restrict__ queries, szs_string_ref_t const *__restrict__ candidates, u32 candidates_count,
    u32 candidate_blocks, u64 *__restrict__ results, u64 results_row_stride, int symmetric, szs_ref_guard_t guard, u32 alphabet,
    u32 blocks_per_group) {
    myers_short_body<runes_, true>(queries, candidates, candidates_count, candidate_blocks, results, results_row_stride, symmetric, guard,
                                   alphabet, blocks_per_group);
}

/* ---- byte queries of more than 2048 bytes: horizontal strips of 2048 rows --------------------------------------------
 *
 *  The reference's serial Myers is block-based and takes any length (serial.hpp:2223-2300); the 64-word register file of
 *  a lane does not.  So a long pattern is cut into STRIPS of up to 64 words.  Inside a strip the add's carry ripples through
 *  all 64 words as in every kernel above; BETWEEN strips only the horizontal deltas of the strip's last row cross - one
 *  (+1, -1) bit pair per text column, Hyyro's block boundary (`Eq |= hn_in`, the shifted-in bits of HP / HN) - and they
 *  are parked, 16 columns to a dword, in a per-workgroup array [dword][lane] in global memory, overwritten in place by
 *  the next strip.  Phantom rows pad the FIRST strip (low bits), so every later strip is full and the pattern's last
 *  row is bit 31 of the last word of the last strip.  distance = len(text) + sum over strips popcount(VP) - popcount(VN).
 *  Persistent grid (the parked arrays are per resident workgroup); work items (query, candidate block), heaviest first.
 */
constexpr int banded_widest_k = 64;            // words of the widest strip: the register file of a lane
constexpr size_t banded_header_bytes_k = 256;

/**
 *  All strips of one (query, candidate block) item at a fixed strip width; returns this lane's sum over the strips of
 *  popcount(VP) - popcount(VN).  Called by the whole workgroup (it synchronises around every Peq build).
 */
template <int words_>
__device__ __forceinline__ i32 myers_strips(u32 *peq, szs_string_ref_t const query, u32 strips, text_stream_t const &text,
                                            u32 text_length, u32 shortest_in_wave, u32 longest_in_wave, u32 *parked_mine) {
    using layout = peq_layout<words_>;
    constexpr u32 strip_rows = 32u * words_;
    u32 const query_length = query.length;
    u32 const pad = strips * strip_rows - query_length; // phantom low rows of the FIRST strip
    u8 const *const pattern = reinterpret_cast<u8 const *>(query.address);
    i32 delta_sum = 0;
    for (u32 strip = 0; strip < strips; ++strip) {
        bool const first_strip = strip == 0, last_strip = strip + 1 == strips;
        // ---- Peq of this strip: bit b of the strip is pattern[strip * strip_rows + b - pad]
        __syncthreads();
        for (int i = threadIdx.x; i < layout::total_dwords; i += 256) peq[i] = 0;
        __syncthreads();
        for (u32 bit = threadIdx.x; bit < strip_rows; bit += 256) {
            u32 const position = strip * strip_rows + bit;
            if (position >= pad) atomicOr(&peq[layout::dword_index(pattern[position - pad], (int)(bit >> 5))], 1u << (bit & 31));
        }
        __syncthreads();

        u32 vp[words_], vn[words_];
#pragma unroll
        for (int w = 0; w < words_; ++w) {
            u32 const first_bit = 32u * w; // phantom rows (first strip only) hold VP = VN = 0
            vp[w] = !first_strip || first_bit >= pad ? ~0u : (first_bit + 32u <= pad ? 0u : (~0u << (pad - first_bit)));
            vn[w] = 0;
        }

        // Deltas of 16 columns per dword: `entering` was parked by the strip above, `leaving` collects this strip's.
        u32 entering = 0, leaving = 0;
        auto take = [&](u32 symbol, u32 column) {
            u32 eq[words_];
            load_match_masks<words_, byte_rows_k>(peq, symbol, eq);
            u32 const slot = 2 * (column & 15u);
            u32 const hp_in = first_strip ? 1u : (entering >> slot) & 1u; // DP row 0 grows by one per column
            u32 const hn_in = first_strip ? 0u : (entering >> (slot + 1)) & 1u;
            leaving |= myers_strip_column<words_>(vp, vn, eq, hp_in, hn_in) << slot;
        };

        u32 column = 0, dword = 0;
        u32 raw_low = text.raw(0);
        // ---- main loop: one text dword (4 columns) per iteration while EVERY live lane still has 4 columns
        if (4 <= shortest_in_wave && longest_in_wave) {
            u32 ahead = text.raw_clamped(1);
            for (; column + 4 <= shortest_in_wave; column += 4, ++dword) {
                u32 const symbols = text.splice(raw_low, ahead);
                raw_low = ahead;
                ahead = text.raw_clamped(dword + 2);
                if ((dword & 3u) == 0) { // a new group of 16 columns
                    if (!first_strip) entering = parked_mine[(u64)(dword / 4) * 256];
                    leaving = 0;
                }
#pragma unroll 1 // one column body per width instead of four: that is what lets all five widths fit 256 registers
                for (int step = 0; step < 4; ++step) take((symbols >> (8 * step)) & 0xFFu, column + step);
                if ((dword & 3u) == 3 && !last_strip) parked_mine[(u64)(dword / 4) * 256] = leaving;
            }
        }
        // A main loop that stops inside a group of 16 columns parks what it has: a lane whose text ends right there
        // never gets to the tail, and a lane that does overwrites the dword with a superset of these bits.
        if ((dword & 3u) != 0 && !last_strip) parked_mine[(u64)(dword / 4) * 256] = leaving;
        // ---- ragged tail: every column predicated on this lane's own length
        if (column < longest_in_wave) {
            u32 next = text.raw(dword + 1);
#pragma unroll 1
            for (; column < longest_in_wave; column += 4, ++dword) {
                u32 const after = text.raw(dword + 2);
                u32 const symbols = text.splice(raw_low, next);
                raw_low = next, next = after;
                if ((dword & 3u) == 0) {
                    if (!first_strip && column < text_length) entering = parked_mine[(u64)(dword / 4) * 256];
                    leaving = 0;
                }
#pragma unroll 1
                for (int step = 0; step < 4; ++step)
                    if (column + step < text_length) take((symbols >> (8 * step)) & 0xFFu, column + step);
                // a group is parked when it is complete or when the text ends inside it
                if (!last_strip && column < text_length && ((dword & 3u) == 3 || column + 4 >= text_length))
                    parked_mine[(u64)(dword / 4) * 256] = leaving;
            }
        }
#pragma unroll
        for (int w = 0; w < words_; ++w) delta_sum += (i32)__builtin_popcount(vp[w]) - (i32)__builtin_popcount(vn[w]);
    }
    return delta_sum;
}

__global__ __launch_bounds__(256, 2) /* two wavefronts per SIMD, like the long kernels */ void levenshtein_myers_banded_kernel(szs_string_ref_t const *__restrict__ queries,
                                                                        u32 queries_count,
                                                                        szs_string_ref_t const *__restrict__ candidates,
                                                                        u32 candidates_count, u32 candidate_blocks,
                                                                        u64 *__restrict__ results, u64 results_row_stride,
                                                                        int symmetric, u32 *__restrict__ parked,
                                                                        u32 parked_dwords, u32 *__restrict__ work_counter) {
    __shared__ __attribute__((aligned(16))) u32 peq[peq_layout<banded_widest_k>::total_dwords];
    __shared__ u32 claimed_work;
    // parked deltas of this workgroup: [dword = 16 columns][lane]
    u32 *const parked_mine = parked + (u64)blockIdx.x * parked_dwords * 256 + threadIdx.x;

    u32 const work_items = queries_count * candidate_blocks;
    for (;;) {
        __syncthreads(); // the previous item's Peq and ticket are no longer in use
        if (threadIdx.x == 0) claimed_work = atomicAdd(work_counter, 1u);
        __syncthreads();
        u32 const work = claimed_work;
        if (work >= work_items) break;
        szs_string_ref_t const query = queries[work % queries_count]; // candidate-block-major, heaviest block first
        u32 const candidate_slot = (candidate_blocks - 1 - work / queries_count) * SZS_CANDIDATES_PER_WORKGROUP + threadIdx.x;
        bool live = candidate_slot < candidates_count;
        szs_string_ref_t candidate = {0, 0, 0};
        if (live) candidate = candidates[candidate_slot];
        if ((symmetric & SZS_LAYOUT_SYMMETRIC) && candidate.index > query.index) live = false;
        u32 const text_length = live ? candidate.length : 0;
        u32 const longest_in_wave = wave_max_u32(text_length);
        u32 const shortest_in_wave = ~wave_max_u32(live ? ~text_length : 0u);
        u64 const safe_address = text_length ? candidate.address : query.address; // lanes without a text: see myers_workgroup
        text_stream_t text(safe_address, text_length);
        if (!text_length) text.valid_dwords = query.length ? 1 : 0;

        // As few strips as the widest width allows, all of the same width, that width rounded up to 8 words: a query of
        // 2150 bytes is two strips of 40 words, not one of 64 and one nearly empty.
        u32 const total_words = __builtin_amdgcn_readfirstlane(query.length ? (query.length + 31u) / 32u : 1u);
        u32 const strips = (total_words + banded_widest_k - 1) / banded_widest_k;
        u32 const strip_words = ((total_words + strips - 1) / strips + 7u) / 8u * 8u;
        i32 delta_sum;
        switch (strip_words) {
        case 8: case 16: case 24: case 32: // (the host sends queries of more than 64 words, i.e. 33 or more words per strip)
            delta_sum = myers_strips<32>(peq, query, strips, text, text_length, shortest_in_wave, longest_in_wave, parked_mine);
            break;
        case 40: delta_sum = myers_strips<40>(peq, query, strips, text, text_length, shortest_in_wave, longest_in_wave, parked_mine); break;
        case 48: delta_sum = myers_strips<48>(peq, query, strips, text, text_length, shortest_in_wave, longest_in_wave, parked_mine); break;
        case 56: delta_sum = myers_strips<56>(peq, query, strips, text, text_length, shortest_in_wave, longest_in_wave, parked_mine); break;
        default: delta_sum = myers_strips<64>(peq, query, strips, text, text_length, shortest_in_wave, longest_in_wave, parked_mine); break;
        }

        if (live) {
            u64 const distance = (u64)((i64)text_length + delta_sum);
            bool const transposed = (symmetric & SZS_LAYOUT_TRANSPOSED) != 0;
            u64 const row = transposed ? candidate.index : query.index, column_of = transposed ? query.index : candidate.index;
            results[row * results_row_stride + column_of] = distance;
            if ((symmetric & SZS_LAYOUT_SYMMETRIC) && candidate.index != query.index)
                results[column_of * results_row_stride + row] = distance;
        }
    }
}

/* ---- long byte queries, SEVERAL LANES PER PAIR -------------------------------------------------------------------------------
 *
 *  One lane per pair makes the longest pair the floor of a launch: a 2048 x 2048-byte pair is 2048 columns x 64 words x 10.5
 *  instructions in ONE lane, 2-5 ms, however idle the rest of the chip is (a wavefront that has its SIMD to itself issues a
 *  dependent instruction every ~8 cycles).  On one GPU the floor hides behind other work; an eighth of config 5 on each of 8
 *  GPUs is 1.7 ms of work under a 5.5 ms floor (profiles/r02/shard_preview.jsonl).
 *
 *  Here a pair is spread over L = 2 or 4 ADJACENT lanes: lane k holds words [k W / L, (k + 1) W / L) of the pattern's
 *  bit-vector and runs k columns behind lane k - 1 - a systolic strip pipeline inside the wavefront.  Between the strips of
 *  neighbouring lanes only the horizontal deltas of the strip's last row cross (Hyyro's block boundary, as between the strips
 *  of the kernel above); they travel with the text symbol in one `v_mov_b32_dpp row_shr:1` per column.  Same instructions per
 *  cell (+2 % for the hand-over), 1 / L of the floor, W / L words of state per lane (no spills at 64 words).
 */
#ifndef SZS_SPLIT_WIDE_BLOCKS
#define SZS_SPLIT_WIDE_BLOCKS 0 /* 1: 256 pairs (256 x L threads) per workgroup - measured slower on config 5 (11.9 vs 11.4 ms:
                                   coarser workgroups, and the other widths' launches already fill the SIMDs); 0: 256 threads */
#endif
template <int lanes_>
constexpr u32 split_threads_k = SZS_SPLIT_WIDE_BLOCKS ? 256u * lanes_ : 256u;

template <int words_per_lane_, int lanes_>
__global__ __launch_bounds__(split_threads_k<lanes_>) void levenshtein_myers_split_kernel(szs_string_ref_t const *__restrict__ queries,
                                                                        szs_string_ref_t const *__restrict__ candidates,
                                                                        u32 candidates_count, u32 candidate_blocks,
                                                                        u64 *__restrict__ results, u64 results_row_stride,
                                                                        int symmetric, szs_ref_guard_t guard) {
    constexpr int words = words_per_lane_ * lanes_;
    constexpr u32 threads = split_threads_k<lanes_>, pairs_per_block = threads / lanes_;
    constexpr int chunks_per_lane = words_per_lane_ / 4;
    static_assert(words_per_lane_ % 4 == 0 && (lanes_ == 2 || lanes_ == 4 || lanes_ == 8), "whole 16-byte Peq chunks per lane");
    using layout = peq_layout<words>;
    __shared__ __attribute__((aligned(16))) u32 peq[layout::total_dwords];

    u32 query_slot, candidate_block;
    myers_work_item(candidate_blocks, query_slot, candidate_block);
    szs_string_ref_t const query = queries[query_slot];
    if (guard.enabled && !ref_is_current(guard, 0, query)) {
        if (threadIdx.x == 0) *guard.stale = guard.sequence;
        return;
    }
    u32 const query_length = query.length;
    u32 const pad = 32u * words - query_length; // phantom low rows: all in lane 0's words (a launch variant spans < 512 rows)
    for (u32 i = threadIdx.x; i < (u32)layout::total_dwords; i += threads) peq[i] = 0;
    __syncthreads();
    {
        u8 const *pattern = reinterpret_cast<u8 const *>(query.address);
        for (u32 i = threadIdx.x; i < query_length; i += threads) {
            u32 const position = pad + i;
            atomicOr(&peq[layout::dword_index(pattern[i], (int)(position >> 5))], 1u << (position & 31));
        }
    }
    __syncthreads();

    u32 const part = threadIdx.x % lanes_, pair = threadIdx.x / lanes_;
    u32 const candidate_slot = candidate_block * pairs_per_block + pair;
    bool live = candidate_slot < candidates_count;
    szs_string_ref_t candidate = {0, 0, 0};
    if (live) candidate = candidates[candidate_slot];
    if (live && guard.enabled && !ref_is_current(guard, 1, candidate)) {
        *guard.stale = guard.sequence;
        live = false;
    }
    if ((symmetric & SZS_LAYOUT_SYMMETRIC) && candidate.index > query.index) live = false;
    u32 const text_length = live ? candidate.length : 0;
    u32 const longest_in_wave = wave_max_u32(text_length);

    u32 vp[words_per_lane_], vn[words_per_lane_];
#pragma unroll
    for (int w = 0; w < words_per_lane_; ++w) {
        u32 const first_bit = 32u * (part * words_per_lane_ + w);
        vp[w] = first_bit >= pad ? ~0u : (first_bit + 32u <= pad ? 0u : (~0u << (pad - first_bit)));
        vn[w] = 0;
    }
    // this lane's Peq chunks: rows of 16 bytes, chunk-major (peq_layout): chunk c of symbol s at uint4 index c * 256 + s
    uint4 const *const my_rows = reinterpret_cast<uint4 const *>(peq) + part * chunks_per_lane * byte_rows_k;

    u64 const safe_address = text_length ? candidate.address : query.address; // lanes without a text: see myers_workgroup
    text_stream_t text(safe_address, text_length);
    if (!text_length) text.valid_dwords = query_length ? 1 : 0;
    u32 raw_low = text.raw(0), next = text.raw(1);
    u32 incoming = 0; // from the lane below: symbol (8 bits) | hp << 8 | hn << 9 | valid << 10 of the column it has just finished
    u32 const steps = longest_in_wave ? longest_in_wave + lanes_ - 1 : 0;
#pragma unroll 1
    for (u32 step = 0, dword = 0; step < steps; step += 4, ++dword) {
        u32 const symbols = text.splice(raw_low, next);
        raw_low = next, next = text.raw(dword + 2);
#pragma unroll
        for (u32 sub = 0; sub < 4; ++sub) {
            bool const head = part == 0;
            u32 const symbol = head ? (symbols >> (8 * sub)) & 0xFFu : incoming & 0xFFu;
            u32 const hp_in = head ? 1u : (incoming >> 8) & 1u; // DP row 0 grows by one per column
            u32 const hn_in = head ? 0u : (incoming >> 9) & 1u;
            bool const active = head ? step + sub < text_length : ((incoming >> 10) & 1u) != 0;
            u32 outgoing = 0;
            if (active) {
                u32 eq[words_per_lane_];
#pragma unroll
                for (int chunk = 0; chunk < chunks_per_lane; ++chunk) {
                    uint4 const row = my_rows[chunk * byte_rows_k + symbol];
                    eq[chunk * 4 + 0] = row.x, eq[chunk * 4 + 1] = row.y, eq[chunk * 4 + 2] = row.z, eq[chunk * 4 + 3] = row.w;
                }
                outgoing = symbol | (myers_strip_column<words_per_lane_>(vp, vn, eq, hp_in, hn_in) << 8) | (1u << 10);
            }
            // row_shr:1 - every lane takes its lower neighbour's word; the first lane of a row of 16 (a `head`) takes zero
            incoming = (u32)__builtin_amdgcn_update_dpp(0, (int)outgoing, 0x111, 0xF, 0xF, true);
        }
    }

    i32 delta = 0;
#pragma unroll
    for (int w = 0; w < words_per_lane_; ++w) delta += (i32)__builtin_popcount(vp[w]) - (i32)__builtin_popcount(vn[w]);
#pragma unroll
    for (int offset = 1; offset < lanes_; offset <<= 1) delta += __shfl_xor(delta, offset, 64);
    if (live && part == 0) {
        u64 const distance = (u64)((i64)text_length + delta);
        bool const transposed = (symmetric & SZS_LAYOUT_TRANSPOSED) != 0;
        u64 const row = transposed ? candidate.index : query.index, column_of = transposed ? query.index : candidate.index;
        results[row * results_row_stride + column_of] = distance;
        if ((symmetric & SZS_LAYOUT_SYMMETRIC) && candidate.index != query.index) results[column_of * results_row_stride + row] = distance;
    }
}

template <int words_per_lane_, int lanes_>
static int launch_myers_split(szs_string_ref_t const *queries, u32 queries_count, szs_string_ref_t const *candidates, u32 candidates_count,
                              u64 *results, u64 stride, int symmetric, szs_ref_guard_t const *guard_or_null, hipStream_t stream) {
    szs_ref_guard_t guard = {};
    if (guard_or_null) guard = *guard_or_null;
    u32 const pairs_per_block = split_threads_k<lanes_> / lanes_;
    u32 const candidate_blocks = (candidates_count + pairs_per_block - 1) / pairs_per_block;
    u32 const queries_per_launch = candidate_blocks ? (1u << 30) / candidate_blocks : queries_count;
    for (u32 first = 0; first < queries_count; first += queries_per_launch) {
        u32 const batch = queries_count - first < queries_per_launch ? queries_count - first : queries_per_launch;
        hipLaunchKernelGGL((levenshtein_myers_split_kernel<words_per_lane_, lanes_>), dim3(batch * candidate_blocks),
                           dim3(split_threads_k<lanes_>), 0, stream, queries + first, candidates, candidates_count, candidate_blocks, results, stride,
                           symmetric, guard);
        hipError_t const error = hipGetLastError();
        if (error != hipSuccess) return (int)error;
    }
    return 0;
}

static u32 banded_grid(u64 work_items) {
    static int resident_of[device_slots_k]; // per device ordinal
    int *const slot = &resident_of[device_slot()];
    int resident = cached(slot);
    if (!resident) {
        int device = 0, units = 0, per_unit = 0;
        if (hipGetDevice(&device) != hipSuccess ||
            hipDeviceGetAttribute(&units, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess ||
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_unit, levenshtein_myers_banded_kernel, 256, 0) != hipSuccess ||
            units <= 0 || per_unit <= 0) {
            (void)hipGetLastError();
            units = 256, per_unit = 1;
        }
        resident = units * per_unit;
        remember(slot, resident);
    }
    return (u32)(work_items < (u64)resident ? work_items : (u64)resident);
}

/* ---- codepoint queries of more than 256 runes ----------------------------------------------------------------------
 *
 *  The short rune kernel keys Peq by the SLOT of a 512-slot rune table - fine for at most 256 distinct runes and 8 words.
 *  A query of 2048 runes may hold 2048 distinct ones; a dense row of 64 words for each would be half a megabyte, and even
 *  the 700 rows that fit a CU's 160 KB leave room for ONE workgroup per CU (config 5u spent 11 ms under its 48-word launch
 *  that way, for 1.7 ms of instructions).  But a pattern of n runes sets n bits in the whole table, so at most n of its
 *  16-byte CHUNKS (4 words, 128 pattern rows) are non-zero, whatever the alphabet.  rune_masks_t stores exactly those:
 *
 *    entries[slots]            hash table, one dword per slot: rune << 11 | id - one LDS read answers a probe;
 *                              ids are dense, 1 .. capacity in the order slots were claimed, 0 = not in the pattern,
 *                              2047 = OVERFLOW (more distinct runes than `capacity`)
 *    pointers[id][chunk]       u16 byte offset of that chunk in the pool; 0 = the all-zero chunk
 *    pool[(32 words + 1) x 16 B]  the non-zero chunks, handed out in the order they were first needed
 *
 *  48 words: 24 KB of pool + 16 KB of hash + 32 B per id - 1,170 ids in half a CU's LDS, where dense rows gave 290.
 *  A column costs one more dependent LDS read (pointers, then chunks) and a `v_bfe` per chunk.  Overflowing runes have their
 *  mask rebuilt from the pattern itself, column by column, only by the lanes that meet one - slow (a pass over the pattern
 *  per column) but exact, rare, and nobody else's business: no failure flag, no second launch.
 */
constexpr u32 rune_id_bits_k = 11, rune_sparse_overflow_k = (1u << rune_id_bits_k) - 1;

template <int words_, int lanes_>
struct rune_masks_t {
    static constexpr int chunks = (words_ + 3) / 4;
    static_assert(chunks % lanes_ == 0, "whole chunks per lane");
    static constexpr int part_chunks = chunks / lanes_;            // chunks one lane of a pair reads per column
    static constexpr int part_stride = (part_chunks + 3) / 4 * 4;  // u16 entries: every part starts 8-byte aligned
    static constexpr int row_stride = part_stride * lanes_;
    static constexpr u32 pool_dwords = (32u * words_ + 1u) * 4u;

    u32 *pool, *entries, *bitmap;
    uint16_t *pointers;
    u32 rune_slots, slot_mask, hash_shift, id_capacity;
    u32 alphabet; // 0: `entries` is the hash table above; else the batch was renumbered 1 ... alphabet (utf8.hip) and
                  // `entries[symbol]` is the symbol's id in THIS pattern, directly (0: not in it)

    /** Dwords of the `entries` region (kept even: the pointers behind it are read 8 bytes at a time). */
    __host__ __device__ static constexpr size_t entry_dwords(size_t slots, size_t alphabet_size) {
        return alphabet_size ? (alphabet_size + 2) / 2 * 2 : slots;
    }
    __device__ __forceinline__ rune_masks_t(u32 *lds, u32 slots, u32 capacity, u32 alphabet_size)
        : pool(lds), entries(lds + pool_dwords), rune_slots(slots), slot_mask(slots - 1), hash_shift(32u - (u32)__builtin_ctz(slots)),
          id_capacity(capacity), alphabet(alphabet_size) {
        pointers = reinterpret_cast<uint16_t *>(entries + entry_dwords(slots, alphabet_size));
        bitmap = reinterpret_cast<u32 *>(pointers + (size_t)(capacity + 1) * row_stride);
    }
    __host__ __device__ static constexpr size_t bytes(size_t slots, size_t capacity, size_t alphabet_size) {
        return ((size_t)pool_dwords + entry_dwords(slots, alphabet_size)) * 4 + (capacity + 1) * row_stride * 2 +
               (((capacity + 1) * chunks + 31) / 32) * 4;
    }
    __device__ __forceinline__ u32 pointer_index(u32 id, u32 chunk) const {
        return id * row_stride + (chunk / part_chunks) * part_stride + chunk % part_chunks;
    }
    __device__ __forceinline__ u32 id_of(u32 rune) const { // 0: not in the pattern
        if (alphabet) return entries[rune];
        u32 slot = (rune * 2654435761u) >> hash_shift;
        for (;;) {
            u32 const entry = entries[slot];
            if ((entry >> rune_id_bits_k) == rune) return entry & rune_sparse_overflow_k;
            if (entry == rune_slot_empty_k) return 0;
            slot = (slot + 1) & slot_mask;
        }
    }

    /** All `threads` threads of the workgroup; `counters` are two dwords of static LDS. */
    __device__ __forceinline__ void build(u32 const *pattern, u32 length, u32 pad, u32 threads, u32 *counters) const {
        for (u32 i = threadIdx.x; i < (length + 1) * 4; i += threads) pool[i] = 0;
        if (alphabet)
            for (u32 i = threadIdx.x; i <= alphabet; i += threads) entries[i] = 0;
        else
            for (u32 i = threadIdx.x; i < rune_slots; i += threads) entries[i] = rune_slot_empty_k;
        u32 const pointer_dwords = (id_capacity + 1) * row_stride / 2, bitmap_dwords = ((id_capacity + 1) * chunks + 31) / 32;
        for (u32 i = threadIdx.x; i < pointer_dwords; i += threads) reinterpret_cast<u32 *>(pointers)[i] = 0;
        for (u32 i = threadIdx.x; i < bitmap_dwords; i += threads) bitmap[i] = 0;
        if (threadIdx.x < 2) counters[threadIdx.x] = 0;
        __syncthreads();
        if (alphabet) { // the first thread to meet a symbol numbers it
            for (u32 i = threadIdx.x; i < length; i += threads) {
                u32 const symbol = pattern[i];
                if (atomicCAS(&entries[symbol], 0u, ~0u) == 0u) {
                    u32 const id = atomicAdd(&counters[0], 1u) + 1;
                    entries[symbol] = id <= id_capacity ? id : rune_sparse_overflow_k;
                }
            }
        }
        else {
        for (u32 i = threadIdx.x; i < length; i += threads) { // claim a slot per distinct rune: at most 32 words_ runes in twice as many slots
            u32 const rune = pattern[i], key = rune << rune_id_bits_k;
            u32 slot = (rune * 2654435761u) >> hash_shift;
            for (;;) {
                u32 const previous = atomicCAS(&entries[slot], rune_slot_empty_k, key);
                if (previous == rune_slot_empty_k || (previous >> rune_id_bits_k) == rune) break;
                slot = (slot + 1) & slot_mask;
            }
        }
        __syncthreads();
        for (u32 slot = threadIdx.x; slot < rune_slots; slot += threads)
            if (entries[slot] != rune_slot_empty_k) {
                u32 const id = atomicAdd(&counters[0], 1u) + 1;
                entries[slot] |= id <= id_capacity ? id : rune_sparse_overflow_k;
            }
        }
        __syncthreads();
        for (u32 i = threadIdx.x; i < length; i += threads) { // the first to need (id, chunk) takes the next chunk of the pool
            u32 const id = id_of(pattern[i]), chunk = (pad + i) >> 7;
            if (id == rune_sparse_overflow_k) continue;
            u32 const bit = id * chunks + chunk, mask = 1u << (bit & 31);
            if (!(atomicOr(&bitmap[bit >> 5], mask) & mask))
                pointers[pointer_index(id, chunk)] = (uint16_t)((atomicAdd(&counters[1], 1u) + 1) * 16);
        }
        __syncthreads();
        for (u32 i = threadIdx.x; i < length; i += threads) {
            u32 const id = id_of(pattern[i]), position = pad + i;
            if (id == rune_sparse_overflow_k) continue;
            atomicOr(&pool[pointers[pointer_index(id, position >> 7)] / 4 + ((position >> 5) & 3)], 1u << (position & 31));
        }
        __syncthreads();
    }

    /** The words [part x part_chunks x 4, ...) of rune `id`'s mask (`id` is not the overflow id). */
    template <int count_>
    __device__ __forceinline__ void load(u32 id, u32 part, u32 (&eq)[count_]) const {
        uint2 const *const row = reinterpret_cast<uint2 const *>(pointers + id * row_stride + part * part_stride);
        char const *const base = reinterpret_cast<char const *>(pool);
#pragma unroll
        for (int group = 0; group < part_stride / 4; ++group) {
            uint2 const four = row[group];
            u32 const offsets[4] = {four.x & 0xFFFFu, four.x >> 16, four.y & 0xFFFFu, four.y >> 16};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                int const chunk = group * 4 + k;
                if (chunk >= part_chunks) continue;
                uint4 const words = *reinterpret_cast<uint4 const *>(base + offsets[k]);
                if (chunk * 4 + 0 < count_) eq[chunk * 4 + 0] = words.x;
                if (chunk * 4 + 1 < count_) eq[chunk * 4 + 1] = words.y;
                if (chunk * 4 + 2 < count_) eq[chunk * 4 + 2] = words.z;
                if (chunk * 4 + 3 < count_) eq[chunk * 4 + 3] = words.w;
            }
        }
    }
};

template <int words_>
__device__ __forceinline__ void myers_long_runes_workgroup(u32 *lds, u32 rune_slots, u32 id_capacity, u32 alphabet, szs_string_ref_t const query,
                                                           szs_string_ref_t const *__restrict__ candidates,
                                                           u32 candidates_count, u32 candidate_block,
                                                           u64 *__restrict__ results, u64 results_row_stride, int symmetric) {
    rune_masks_t<words_, 1> const masks(lds, rune_slots, id_capacity, alphabet);
    __shared__ u32 counters[2];
    u32 const query_length = query.length;
    u32 const pad = 32u * words_ - query_length; // phantom low rows
    u32 const *const pattern = reinterpret_cast<u32 const *>(query.address);
    masks.build(pattern, query_length, pad, 256, counters);

    // ---- this lane's candidate
    u32 const candidate_slot = candidate_block * SZS_CANDIDATES_PER_WORKGROUP + threadIdx.x;
    bool live = candidate_slot < candidates_count;
    szs_string_ref_t candidate = {0, 0, 0};
    if (live) candidate = candidates[candidate_slot];
    if ((symmetric & SZS_LAYOUT_SYMMETRIC) && candidate.index > query.index) live = false;
    u32 const text_length = live ? candidate.length : 0;
    u32 const longest_in_wave = wave_max_u32(text_length);
    u32 const shortest_in_wave = ~wave_max_u32(live ? ~text_length : 0u);

    u32 vp[words_], vn[words_];
#pragma unroll
    for (int w = 0; w < words_; ++w) {
        u32 const first_bit = 32u * w;
        vp[w] = first_bit >= pad ? ~0u : (first_bit + 32u <= pad ? 0u : (~0u << (pad - first_bit)));
        vn[w] = 0;
    }

    auto take = [&](u32 rune) {
        u32 eq[words_];
        u32 const id = masks.id_of(rune);
        if (id != rune_sparse_overflow_k) masks.load(id, 0, eq);
        else { // a rune past the table's capacity: its mask, straight from the pattern
#pragma unroll
            for (int w = 0; w < words_; ++w) {
                u32 bits = 0;
#pragma unroll 1
                for (u32 bit = 0; bit < 32; ++bit) {
                    u32 const position = 32u * w + bit;
                    if (position >= pad && pattern[position - pad] == rune) bits |= 1u << bit;
                }
                eq[w] = bits;
            }
        }
        myers_column<words_>(vp, vn, eq);
    };

    // ---- codepoints: four columns per 16-byte load, loaded one iteration early (see myers_workgroup)
    u32 const *const runes = reinterpret_cast<u32 const *>(candidate.address);
    auto quad_at = [&](u32 index) -> uint4 {
        return index < text_length ? *reinterpret_cast<uint4 const *>(runes + index) : make_uint4(0, 0, 0, 0);
    };
    u32 column = 0;
    if (4 <= shortest_in_wave && longest_in_wave) {
        uint4 ahead = quad_at(0);
        for (; column + 4 <= shortest_in_wave; column += 4) {
            uint4 const now = ahead;
            ahead = quad_at(column + 4);
            take(now.x), take(now.y), take(now.z), take(now.w);
        }
    }
#pragma unroll 1
    for (; column < longest_in_wave; ++column)
        if (column < text_length) take(runes[column]);

    if (live) {
        u32 distance = text_length;
#pragma unroll
        for (int w = 0; w < words_; ++w) distance += (u32)__builtin_popcount(vp[w]) - (u32)__builtin_popcount(vn[w]);
        bool const transposed = (symmetric & SZS_LAYOUT_TRANSPOSED) != 0;
        u64 const row = transposed ? candidate.index : query.index, column_of = transposed ? query.index : candidate.index;
        results[row * results_row_stride + column_of] = distance;
        if ((symmetric & SZS_LAYOUT_SYMMETRIC) && candidate.index != query.index)
            results[column_of * results_row_stride + row] = distance;
    }
}

template <int words_>
__global__ __launch_bounds__(256) void levenshtein_myers_long_runes_kernel(szs_string_ref_t const *__restrict__ queries,
                                                                            szs_string_ref_t const *__restrict__ candidates,
                                                                            u32 candidates_count, u32 candidate_blocks,
                                                                            u64 *__restrict__ results, u64 results_row_stride,
                                                                            int symmetric, u32 rune_slots, u32 id_capacity, u32 alphabet) {
    extern __shared__ __attribute__((aligned(16))) u32 rune_lds[];
    u32 query_slot, candidate_block;
    myers_work_item(candidate_blocks, query_slot, candidate_block);
    myers_long_runes_workgroup<words_>(rune_lds, rune_slots, id_capacity, alphabet, queries[query_slot], candidates, candidates_count,
                                       candidate_block, results, results_row_stride, symmetric);
}

/** LDS plan of the strip kernel (dense rows per rune id, u16 ids beside u32 keys): slots = twice the runes a query of `words` words can hold; as many Peq rows as the
 *  budget leaves - two workgroups per CU (80 KB each) when that still numbers 512 runes, else one (all 160 KB). */
static bool rune_lds_plan_dense(unsigned words, u32 &rune_slots, u32 &id_capacity, size_t &bytes) {
    rune_slots = 1;
    while (rune_slots < 64u * words) rune_slots *= 2;
    size_t const table_bytes = (size_t)rune_slots * (sizeof(u32) + sizeof(uint16_t)), row_bytes = (size_t)((words + 3) / 4) * 16;
    size_t const most_runes = 32u * words;
    long const forced = szs_tuning_get(szs_knob_rune_ids_k); // a testing aid: shrinks the table so that runes overflow it
    size_t const shared = ((size_t)80 << 10) - 1024, whole = ((size_t)160 << 10) - 1024; // a little static LDS on top
    size_t capacity = shared > table_bytes + row_bytes ? (shared - table_bytes) / row_bytes - 1 : 0;
    if (capacity < 512 && capacity < most_runes) capacity = whole > table_bytes + row_bytes ? (whole - table_bytes) / row_bytes - 1 : 0;
    if (capacity > most_runes) capacity = most_runes;
    if (capacity > 0xFFF0) capacity = 0xFFF0;
    if (forced > 0 && (size_t)forced < capacity) capacity = (size_t)forced;
    if (capacity < 1) return false;
    id_capacity = (u32)capacity;
    bytes = (capacity + 1) * row_bytes + table_bytes;
    return true;
}

/**
 *  LDS plan of one long rune launch (rune_masks_t<words, lanes>): slots = twice the runes a query of `words` words can hold;
 *  the SMALLEST share of a CU's 160 KB (a quarter, a third, half, all) whose id capacity covers 768 distinct runes or every
 *  rune the query can hold - text repeats its runes (config 5u's 1560-rune strings hold ~400 distinct ones, 2048 CJK
 *  characters of running text ~600); what overflows takes the slow path, correctly.
 */
template <int words_, int lanes_>
static bool rune_lds_plan(u32 alphabet, u32 &rune_slots, u32 &id_capacity, size_t &bytes) {
    using masks = rune_masks_t<words_, lanes_>;
    rune_slots = 1;
    while (rune_slots < 64u * words_) rune_slots *= 2;
    size_t const most_runes = 32u * words_, wanted = most_runes < 768 ? most_runes : 768;
    long const forced = szs_tuning_get(szs_knob_rune_ids_k); // a testing aid: shrinks the table so that runes overflow it
    size_t capacity = 0;
    for (size_t share = 4; share >= 1; --share) {
        size_t const budget = ((size_t)160 << 10) / share - 1024; // a little static LDS on top
        size_t low = 0, high = rune_sparse_overflow_k - 1;         // the largest capacity whose table fits the budget
        while (low < high) {
            size_t const middle = (low + high + 1) / 2;
            if (masks::bytes(rune_slots, middle, alphabet) <= budget) low = middle;
            else high = middle - 1;
        }
        capacity = masks::bytes(rune_slots, low, alphabet) <= budget ? low : 0;
        if (capacity >= wanted) break;
    }
    if (capacity > most_runes) capacity = most_runes;
    if (forced > 0 && (size_t)forced < capacity) capacity = (size_t)forced;
    if (capacity < 1) return false;
    id_capacity = (u32)capacity;
    bytes = (masks::bytes(rune_slots, capacity, alphabet) + 15) / 16 * 16;
    return true;
}

template <int words_>
static int launch_long_runes(szs_string_ref_t const *queries, u32 queries_count, szs_string_ref_t const *candidates,
                             u32 candidates_count, u64 *results, u64 stride, int symmetric, u32 alphabet, hipStream_t stream) {
    u32 rune_slots = 0, id_capacity = 0;
    size_t bytes = 0;
    if (!rune_lds_plan<words_, 1>(alphabet, rune_slots, id_capacity, bytes)) return (int)hipErrorNotSupported;
    static int granted_on[device_slots_k]; // per width and device: has this much dynamic LDS been granted to the kernel?
    int *const granted = &granted_on[device_slot()];
    if (!cached(granted)) {
        hipError_t const error = hipFuncSetAttribute(reinterpret_cast<void const *>(levenshtein_myers_long_runes_kernel<words_>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)(((size_t)160 << 10) - 1024));
        if (error != hipSuccess) {
            (void)hipGetLastError();
            return (int)hipErrorNotSupported; // the host falls back to the rune-keyed DP kernel
        }
        remember(granted, 1);
    }
    u32 const candidate_blocks = (candidates_count + SZS_CANDIDATES_PER_WORKGROUP - 1) / SZS_CANDIDATES_PER_WORKGROUP;
    u32 const queries_per_launch = candidate_blocks ? (1u << 30) / candidate_blocks : queries_count;
    for (u32 first = 0; first < queries_count; first += queries_per_launch) {
        u32 const batch = queries_count - first < queries_per_launch ? queries_count - first : queries_per_launch;
        hipLaunchKernelGGL(levenshtein_myers_long_runes_kernel<words_>, dim3(batch * candidate_blocks), dim3(256), bytes, stream,
                           queries + first, candidates, candidates_count, candidate_blocks, results, stride, symmetric,
                           rune_slots, id_capacity, alphabet);
        hipError_t const error = hipGetLastError();
        if (error != hipSuccess) return (int)error;
    }
    return 0;
}

/* ---- long codepoint queries, several lanes per pair ----------------------------------------------------------------------
 *
 *  levenshtein_myers_split_kernel for UTF-32 texts.  Here the split buys more than the shorter floor: a 48- or 64-word rune
 *  query needs most of a CU's LDS for its match-mask rows (one per DISTINCT rune, up to 700), so the one-lane-per-pair kernel
 *  above runs ONE workgroup of four wavefronts per CU - a wavefront per SIMD, every LDS round trip exposed (config 5u: the
 *  48-word launch took 8.9 ms for 1.7 ms of instructions, profiles/r02/pmc_configs.json).  This kernel keeps 256 pairs per
 *  workgroup - the same table serves 256 x L threads, L wavefronts per SIMD - and each lane carries 1 / L of the state.
 *
 *  The head lane of a pair resolves the rune's row id (one probe per pair and column, not per lane) and hands it up with the
 *  boundary deltas: id (16 bits) | hp << 16 | hn << 17 | valid << 18.  A rune beyond the table's capacity (id 2047) travels
 *  in a second move, taken only by wavefronts that hold one, and every lane rebuilds its own words of the mask from the pattern.
 */
template <int words_per_lane_, int lanes_>
__global__ __launch_bounds__(256 * lanes_) void levenshtein_myers_split_runes_kernel(szs_string_ref_t const *__restrict__ queries,
                                                                                      szs_string_ref_t const *__restrict__ candidates,
                                                                                      u32 candidates_count, u32 candidate_blocks,
                                                                                      u64 *__restrict__ results, u64 results_row_stride,
                                                                                      int symmetric, u32 rune_slots, u32 id_capacity, u32 alphabet) {
    constexpr int words = words_per_lane_ * lanes_;
    // 256 pairs per workgroup when the launch fills the device (one table serves L wavefronts per SIMD); 64 in a launch of a few
    // workgroups, whose pairs are its duration: sixteen wavefronts on a CU take turns on its SIMDs while other CUs idle
    // (real-text lines: three 32-word queries were 48 workgroups and 3.2 ms - profiles/r03/timeline_lines_utf8_v1.txt)
    u32 const threads = blockDim.x, pairs_per_block = threads / lanes_;
    static_assert(words_per_lane_ % 4 == 0 && (lanes_ == 2 || lanes_ == 4 || lanes_ == 8), "whole 16-byte Peq chunks per lane");
    extern __shared__ __attribute__((aligned(16))) u32 rune_lds[];
    rune_masks_t<words, lanes_> const masks(rune_lds, rune_slots, id_capacity, alphabet);
    __shared__ u32 counters[2];

    u32 query_slot, candidate_block;
    myers_work_item(candidate_blocks, query_slot, candidate_block);
    szs_string_ref_t const query = queries[query_slot];
    u32 const query_length = query.length;
    u32 const pad = 32u * words - query_length; // phantom low rows
    u32 const *const pattern = reinterpret_cast<u32 const *>(query.address);
    masks.build(pattern, query_length, pad, threads, counters);

    u32 const part = threadIdx.x % lanes_, pair = threadIdx.x / lanes_;
    u32 const candidate_slot = candidate_block * pairs_per_block + pair;
    bool live = candidate_slot < candidates_count;
    szs_string_ref_t candidate = {0, 0, 0};
    if (live) candidate = candidates[candidate_slot];
    if ((symmetric & SZS_LAYOUT_SYMMETRIC) && candidate.index > query.index) live = false;
    u32 const text_length = live ? candidate.length : 0;
    u32 const longest_in_wave = wave_max_u32(text_length);
    bool const head = part == 0;

    u32 vp[words_per_lane_], vn[words_per_lane_];
#pragma unroll
    for (int w = 0; w < words_per_lane_; ++w) {
        u32 const first_bit = 32u * (part * words_per_lane_ + w);
        vp[w] = first_bit >= pad ? ~0u : (first_bit + 32u <= pad ? 0u : (~0u << (pad - first_bit)));
        vn[w] = 0;
    }
    u32 const *const runes = reinterpret_cast<u32 const *>(candidate.address);
    auto quad_at = [&](u32 index) -> uint4 { // the head lane's text, four columns per 16-byte load (see myers_workgroup)
        return head && index < text_length ? *reinterpret_cast<uint4 const *>(runes + index) : make_uint4(0, 0, 0, 0);
    };
    uint4 quad = quad_at(0), quad_next = quad_at(4);
    u32 incoming = 0, incoming_rune = 0;
    u32 const steps = longest_in_wave ? longest_in_wave + lanes_ - 1 : 0;
#pragma unroll 1
    for (u32 step = 0; step < steps; ++step) {
        u32 const own = quad.x;
        quad.x = quad.y, quad.y = quad.z, quad.z = quad.w;
        if ((step & 3u) == 3u) quad = quad_next, quad_next = quad_at(step + 5);
        bool const active = head ? step < text_length : ((incoming >> 18) & 1u) != 0;
        u32 const rune = head ? own : incoming_rune;
        u32 id = incoming & 0xFFFFu;
        if (head) id = active ? masks.id_of(own) : 0u;
        u32 const hp_in = head ? 1u : (incoming >> 16) & 1u; // DP row 0 grows by one per column
        u32 const hn_in = head ? 0u : (incoming >> 17) & 1u;
        u32 outgoing = 0;
        if (active) {
            u32 eq[words_per_lane_];
            if (id != rune_sparse_overflow_k) masks.load(id, part, eq);
            else { // a rune past the table's capacity: this lane's words of its mask, straight from the pattern
#pragma unroll
                for (int w = 0; w < words_per_lane_; ++w) {
                    u32 bits = 0;
#pragma unroll 1
                    for (u32 bit = 0; bit < 32; ++bit) {
                        u32 const position = 32u * (part * words_per_lane_ + w) + bit;
                        if (position >= pad && pattern[position - pad] == rune) bits |= 1u << bit;
                    }
                    eq[w] = bits;
                }
            }
            outgoing = id | (myers_strip_column<words_per_lane_>(vp, vn, eq, hp_in, hn_in) << 16) | (1u << 18);
        }
        // row_shr:1 - every lane takes its lower neighbour's word; the first lane of a row of 16 (a `head`) takes zero
        incoming = (u32)__builtin_amdgcn_update_dpp(0, (int)outgoing, 0x111, 0xF, 0xF, true);
        if (__any(active && id == rune_sparse_overflow_k))
            incoming_rune = (u32)__builtin_amdgcn_update_dpp(0, (int)rune, 0x111, 0xF, 0xF, true);
    }

    i32 delta = 0;
#pragma unroll
    for (int w = 0; w < words_per_lane_; ++w) delta += (i32)__builtin_popcount(vp[w]) - (i32)__builtin_popcount(vn[w]);
#pragma unroll
    for (int offset = 1; offset < lanes_; offset <<= 1) delta += __shfl_xor(delta, offset, 64);
    if (live && head) {
        u64 const distance = (u64)((i64)text_length + delta);
        bool const transposed = (symmetric & SZS_LAYOUT_TRANSPOSED) != 0;
        u64 const row = transposed ? candidate.index : query.index, column_of = transposed ? query.index : candidate.index;
        results[row * results_row_stride + column_of] = distance;
        if ((symmetric & SZS_LAYOUT_SYMMETRIC) && candidate.index != query.index) results[column_of * results_row_stride + row] = distance;
    }
}

template <int words_per_lane_, int lanes_>
static int launch_split_runes(szs_string_ref_t const *queries, u32 queries_count, szs_string_ref_t const *candidates, u32 candidates_count,
                              u64 *results, u64 stride, int symmetric, u32 alphabet, u32 pairs_per_block, hipStream_t stream) {
    u32 rune_slots = 0, id_capacity = 0;
    size_t bytes = 0;
    if (!rune_lds_plan<words_per_lane_ * lanes_, lanes_>(alphabet, rune_slots, id_capacity, bytes)) return (int)hipErrorNotSupported;
    static int granted_on[device_slots_k];
    int *const granted = &granted_on[device_slot()];
    if (!cached(granted)) {
        hipError_t const error = hipFuncSetAttribute(reinterpret_cast<void const *>(levenshtein_myers_split_runes_kernel<words_per_lane_, lanes_>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)(((size_t)160 << 10) - 1024));
        if (error != hipSuccess) {
            (void)hipGetLastError();
            return (int)hipErrorNotSupported;
        }
        remember(granted, 1);
    }
    if (pairs_per_block != 64u) pairs_per_block = 256u;
    u32 const candidate_blocks = (candidates_count + pairs_per_block - 1) / pairs_per_block;
    u32 const queries_per_launch = candidate_blocks ? (1u << 30) / candidate_blocks : queries_count;
    for (u32 first = 0; first < queries_count; first += queries_per_launch) {
        u32 const batch = queries_count - first < queries_per_launch ? queries_count - first : queries_per_launch;
        hipLaunchKernelGGL((levenshtein_myers_split_runes_kernel<words_per_lane_, lanes_>), dim3(batch * candidate_blocks), dim3(pairs_per_block * lanes_),
                           bytes, stream, queries + first, candidates, candidates_count, candidate_blocks, results, stride, symmetric, rune_slots,
                           id_capacity, alphabet);
        hipError_t const error = hipGetLastError();
        if (error != hipSuccess) return (int)error;
    }
    return 0;
}

constexpr u32 rune_overflow_id_k = 0xFFFFu; // the strip kernel below: dense rows, u16 ids beside the keys

/* ---- codepoint queries of more than 2048 runes: the strips of the byte kernel with the rune table of the long rune kernels -----
 *
 *  The reference's rune Myers is length-agnostic (serial.hpp:2328-2512: an open-addressing Peq per 64-rune block).  Here a
 *  long pattern is cut into strips of up to 2048 runes exactly like a long byte pattern (myers_strips above): the carry
 *  ripples through a strip's words, the horizontal deltas under its last row are parked per text column, 16 to a dword.
 *  What differs is the match-mask table: a strip's runes get dense ids (claim slots with atomicCAS, number the claimed slots)
 *  and Peq has one row per id, REBUILT FOR EVERY STRIP from that strip's runes only; runes past the table's capacity get
 *  the overflow id and their mask is rebuilt from the strip's slice of the pattern by the lanes that meet one.  Columns
 *  are taken one at a time, each predicated on the lane's own text length: this kernel exists so that no codepoint query
 *  falls back to the cell-by-cell recurrences, not to set records.
 */
template <int words_>
__global__ __launch_bounds__(256) void levenshtein_myers_banded_runes_kernel(
    szs_string_ref_t const *__restrict__ queries, u32 queries_count, szs_string_ref_t const *__restrict__ candidates,
    u32 candidates_count, u32 candidate_blocks, u64 *__restrict__ results, u64 results_row_stride, int symmetric,
    u32 *__restrict__ parked, u32 parked_dwords, u32 *__restrict__ work_counter, u32 rune_slots, u32 id_capacity) {
    extern __shared__ __attribute__((aligned(16))) u32 strip_lds[];
    __shared__ u32 claimed_work, claimed_ids;
    constexpr int chunks = (words_ + 3) / 4;
    constexpr u32 strip_rows = 32u * words_;
    u32 const rows = id_capacity + 1; // row 0: runes the strip does not contain
    u32 *const peq = strip_lds;
    u32 *const keys = peq + (size_t)chunks * rows * 4;
    uint16_t *const ids = reinterpret_cast<uint16_t *>(keys + rune_slots);
    u32 const slot_mask = rune_slots - 1, hash_shift = 32u - (u32)__builtin_ctz(rune_slots);
    auto dword_index = [&](u32 row, u32 w) -> u32 { return ((w / 4) * rows + row) * 4 + (w % 4); };
    u32 *const parked_mine = parked + (u64)blockIdx.x * parked_dwords * 256 + threadIdx.x; // [dword = 16 columns][lane]

    u32 const work_items = queries_count * candidate_blocks;
    for (;;) {
        __syncthreads();
        if (threadIdx.x == 0) claimed_work = atomicAdd(work_counter, 1u);
        __syncthreads();
        u32 const work = claimed_work;
        if (work >= work_items) break;
        szs_string_ref_t const query = queries[work % queries_count]; // candidate-block-major, heaviest block first
        u32 const candidate_slot = (candidate_blocks - 1 - work / queries_count) * SZS_CANDIDATES_PER_WORKGROUP + threadIdx.x;
        bool live = candidate_slot < candidates_count;
        szs_string_ref_t candidate = {0, 0, 0};
        if (live) candidate = candidates[candidate_slot];
        if ((symmetric & SZS_LAYOUT_SYMMETRIC) && candidate.index > query.index) live = false;
        u32 const text_length = live ? candidate.length : 0;
        u32 const longest_in_wave = wave_max_u32(text_length);
        u32 const *const runes = reinterpret_cast<u32 const *>(candidate.address);
        u32 const *const pattern = reinterpret_cast<u32 const *>(query.address);

        u32 const query_length = query.length;
        u32 const total_words = query_length ? (query_length + 31u) / 32u : 1u;
        u32 const strips = (total_words + words_ - 1) / words_;
        u32 const pad = strips * strip_rows - query_length; // phantom low rows of the FIRST strip
        i32 delta_sum = 0;
        for (u32 strip = 0; strip < strips; ++strip) {
            bool const first_strip = strip == 0, last_strip = strip + 1 == strips;
            u32 const strip_base = strip * strip_rows; // bit b of the strip is pattern[strip_base + b - pad]
            // ---- rune table and Peq of this strip
            __syncthreads();
            for (u32 i = threadIdx.x; i < (u32)chunks * rows * 4; i += 256) peq[i] = 0;
            for (u32 i = threadIdx.x; i < rune_slots; i += 256) keys[i] = rune_slot_empty_k, ids[i] = 0;
            if (threadIdx.x == 0) claimed_ids = 0;
            __syncthreads();
            for (u32 bit = threadIdx.x; bit < strip_rows; bit += 256) {
                u32 const position = strip_base + bit;
                if (position < pad) continue;
                u32 const rune = pattern[position - pad];
                u32 slot = (rune * 2654435761u) >> hash_shift;
                for (;;) {
                    u32 const previous = atomicCAS(&keys[slot], rune_slot_empty_k, rune);
                    if (previous == rune_slot_empty_k || previous == rune) break;
                    slot = (slot + 1) & slot_mask;
                }
            }
            __syncthreads();
            for (u32 slot = threadIdx.x; slot < rune_slots; slot += 256)
                if (keys[slot] != rune_slot_empty_k) {
                    u32 const id = atomicAdd(&claimed_ids, 1u) + 1;
                    ids[slot] = (uint16_t)(id <= id_capacity ? id : rune_overflow_id_k);
                }
            __syncthreads();
            auto id_of = [&](u32 rune) -> u32 { // 0: not in this strip
                u32 slot = (rune * 2654435761u) >> hash_shift;
                for (;;) {
                    u32 const key = keys[slot];
                    if (key == rune) return ids[slot];
                    if (key == rune_slot_empty_k) return 0;
                    slot = (slot + 1) & slot_mask;
                }
            };
            for (u32 bit = threadIdx.x; bit < strip_rows; bit += 256) {
                u32 const position = strip_base + bit;
                if (position < pad) continue;
                u32 const id = id_of(pattern[position - pad]);
                if (id != rune_overflow_id_k) atomicOr(&peq[dword_index(id, bit >> 5)], 1u << (bit & 31));
            }
            __syncthreads();

            u32 vp[words_], vn[words_];
#pragma unroll
            for (int w = 0; w < words_; ++w) {
                u32 const first_bit = 32u * w; // phantom rows (first strip only) hold VP = VN = 0
                vp[w] = !first_strip || first_bit >= pad ? ~0u : (first_bit + 32u <= pad ? 0u : (~0u << (pad - first_bit)));
                vn[w] = 0;
            }
            u32 entering = 0, leaving = 0;
#pragma unroll 1
            for (u32 column = 0; column < longest_in_wave; ++column) {
                if (column >= text_length) continue;
                if ((column & 15u) == 0) {
                    entering = first_strip ? 0u : parked_mine[(u64)(column / 16) * 256];
                    leaving = 0;
                }
                u32 const rune = runes[column];
                u32 eq[words_];
                u32 const id = id_of(rune);
                if (id != rune_overflow_id_k) {
                    uint4 const *table = reinterpret_cast<uint4 const *>(peq);
#pragma unroll
                    for (int chunk = 0; chunk < chunks; ++chunk) {
                        uint4 const row = table[chunk * rows + id];
                        if (chunk * 4 + 0 < words_) eq[chunk * 4 + 0] = row.x;
                        if (chunk * 4 + 1 < words_) eq[chunk * 4 + 1] = row.y;
                        if (chunk * 4 + 2 < words_) eq[chunk * 4 + 2] = row.z;
                        if (chunk * 4 + 3 < words_) eq[chunk * 4 + 3] = row.w;
                    }
                }
                else { // a rune past the table's capacity: its mask, straight from this strip's slice of the pattern
#pragma unroll
                    for (int w = 0; w < words_; ++w) {
                        u32 bits = 0;
#pragma unroll 1
                        for (u32 bit = 0; bit < 32; ++bit) {
                            u32 const position = strip_base + 32u * w + bit;
                            if (position >= pad && pattern[position - pad] == rune) bits |= 1u << bit;
                        }
                        eq[w] = bits;
                    }
                }
                u32 const slot = 2 * (column & 15u);
                u32 const hp_in = first_strip ? 1u : (entering >> slot) & 1u; // DP row 0 grows by one per column
                u32 const hn_in = first_strip ? 0u : (entering >> (slot + 1)) & 1u;
                leaving |= myers_strip_column<words_>(vp, vn, eq, hp_in, hn_in) << slot;
                if (!last_strip && ((column & 15u) == 15 || column + 1 == text_length)) parked_mine[(u64)(column / 16) * 256] = leaving;
            }
#pragma unroll
            for (int w = 0; w < words_; ++w) delta_sum += (i32)__builtin_popcount(vp[w]) - (i32)__builtin_popcount(vn[w]);
        }

        if (live) {
            u64 const distance = (u64)((i64)text_length + delta_sum);
            bool const transposed = (symmetric & SZS_LAYOUT_TRANSPOSED) != 0;
            u64 const row = transposed ? candidate.index : query.index, column_of = transposed ? query.index : candidate.index;
            results[row * results_row_stride + column_of] = distance;
            if ((symmetric & SZS_LAYOUT_SYMMETRIC) && candidate.index != query.index) results[column_of * results_row_stride + row] = distance;
        }
    }
}

constexpr int banded_runes_words_k = 64; // one instantiation: strips of 2048 runes (a 2049-rune query is two strips)

/** Resident workgroups of the codepoint strip kernel with `lds_bytes` of dynamic LDS, per device. */
static u32 banded_runes_grid(u64 work_items, size_t lds_bytes) {
    static int resident_of[device_slots_k];
    int *const slot = &resident_of[device_slot()];
    int resident = cached(slot);
    if (!resident) {
        int device = 0, units = 0, per_unit = 0;
        if (hipGetDevice(&device) != hipSuccess ||
            hipDeviceGetAttribute(&units, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess ||
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_unit, levenshtein_myers_banded_runes_kernel<banded_runes_words_k>, 256,
                                                         lds_bytes) != hipSuccess ||
            units <= 0 || per_unit <= 0) {
            (void)hipGetLastError();
            units = 256, per_unit = 1;
        }
        resident = units * per_unit;
        remember(slot, resident);
    }
    return (u32)(work_items < (u64)resident ? work_items : (u64)resident);
}

template <typename kernel_t, typename... extra_t>
static int launch_myers(kernel_t kernel, szs_string_ref_t const *queries, u32 queries_count,
                        szs_string_ref_t const *candidates, u32 candidates_count, u64 *results, u64 stride, int symmetric,
                        szs_ref_guard_t const *guard_or_null, hipStream_t stream, size_t dynamic_lds = 0, extra_t... extra) {
    szs_ref_guard_t guard = {};
    if (guard_or_null) guard = *guard_or_null;
    u32 const candidate_blocks = (candidates_count + SZS_CANDIDATES_PER_WORKGROUP - 1) / SZS_CANDIDATES_PER_WORKGROUP;
    // Keep each grid under 2^30 workgroups; enormous cross-products are cut along the query axis.
    u32 const queries_per_launch = candidate_blocks ? (1u << 30) / candidate_blocks : queries_count;
    for (u32 first = 0; first < queries_count; first += queries_per_launch) {
        u32 const batch = queries_count - first < queries_per_launch ? queries_count - first : queries_per_launch;
        hipLaunchKernelGGL(kernel, dim3(batch * candidate_blocks), dim3(256), dynamic_lds, stream, queries + first, candidates,
                           candidates_count, candidate_blocks, results, stride, symmetric, guard, extra...);
        hipError_t const error = hipGetLastError();
        if (error != hipSuccess) return (int)error;
    }
    return 0;
}

/**
 *  The mixed-width short kernels.  A launch of tens of thousands of workgroups of tiny strings (4096 x 4096 words of text:
 *  65,536 workgroups that live ~5 us, most of it round trips to the query and its table) merges candidate blocks: a workgroup
 *  takes 2, 4 or 8 consecutive blocks as long as 16,384 workgroups remain - config 2 (4096) and config 5 (28,769) never merge.
 */
template <bool runes_>
static int launch_myers_short(szs_string_ref_t const *queries, u32 queries_count, szs_string_ref_t const *candidates,
                              u32 candidates_count, u64 *results, u64 stride, int symmetric, szs_ref_guard_t const *guard_or_null,
                              hipStream_t stream, size_t dynamic_lds, u32 alphabet) {
    szs_ref_guard_t guard = {};
    if (guard_or_null) guard = *guard_or_null;
    u32 const candidate_blocks = (candidates_count + SZS_CANDIDATES_PER_WORKGROUP - 1) / SZS_CANDIDATES_PER_WORKGROUP;
    u32 blocks_per_group = 1;
    int const pinned = szs_tuning_get(szs_knob_merge_k); // a testing aid: blocks per workgroup, 1 = never merge
    if (pinned > 0) blocks_per_group = (u32)pinned < candidate_blocks ? (u32)pinned : candidate_blocks;
    else
        while (blocks_per_group < 8 && (u64)queries_count * ((candidate_blocks + 2 * blocks_per_group - 1) / (2 * blocks_per_group)) >= 16384)
            blocks_per_group *= 2;
    u32 const groups = (candidate_blocks + blocks_per_group - 1) / blocks_per_group;
    u32 const queries_per_launch = groups ? (1u << 30) / groups : queries_count;
    for (u32 first = 0; first < queries_count; first += queries_per_launch) {
        u32 const batch = queries_count - first < queries_per_launch ? queries_count - first : queries_per_launch;
        if (blocks_per_group > 1)
            hipLaunchKernelGGL((levenshtein_myers_short_merged_kernel<runes_>), dim3(batch * groups), dim3(256), dynamic_lds, stream, queries + first,
                               candidates, candidates_count, groups, results, stride, symmetric, guard, alphabet, blocks_per_group);
        else
            hipLaunchKernelGGL((levenshtein_myers_short_kernel<runes_>), dim3(batch * groups), dim3(256), dynamic_lds, stream, queries + first,
                               candidates, candidates_count, groups, results, stride, symmetric, guard, alphabet, 1u);
        hipError_t const error = hipGetLastError();
        if (error != hipSuccess) return (int)error;
    }
    return 0;
}

} // namespace szs_hip

extern "C" int szs_hip_levenshtein_myers(unsigned words, szs_string_ref_t const *queries, uint32_t queries_count,
                                         szs_string_ref_t const *candidates, uint32_t candidates_count,
                                         uint64_t *results, uint64_t results_row_stride, int symmetric,
                                         szs_ref_guard_t const *guard, void *stream) {
    using namespace szs_hip;
    if (!queries_count || !candidates_count) return 0;
    hipStream_t const s = static_cast<hipStream_t>(stream);
#define SZS_MYERS_CASE(W)                                                                                              \
    case W:                                                                                                            \
        return launch_myers(levenshtein_myers_long_kernel<W>, queries, queries_count, candidates, candidates_count,   \
                            results, results_row_stride, symmetric, guard, s);
    switch (words) {
    case SZS_MYERS_SHORT_WORDS:
        return launch_myers_short<false>(queries, queries_count, candidates, candidates_count, results, results_row_stride, symmetric, guard, s, 0,
                                         0u);
        SZS_MYERS_CASE(10)
        SZS_MYERS_CASE(12)
        SZS_MYERS_CASE(16)
        SZS_MYERS_CASE(20)
        SZS_MYERS_CASE(24)
        SZS_MYERS_CASE(32)
        SZS_MYERS_CASE(48)
        SZS_MYERS_CASE(64)
    default: return (int)hipErrorInvalidValue;
    }
#undef SZS_MYERS_CASE
}

extern "C" int szs_hip_levenshtein_myers_fused(szs_fused_plan_t const *plan_of_call, uint64_t *results, uint64_t results_row_stride, int layout,
                                               void *stream) {
    using namespace szs_hip;
    szs_fused_plan_t plan = *plan_of_call;
    plan.symmetric = (layout & SZS_LAYOUT_SYMMETRIC) != 0;
    uint32_t const queries_count = plan.side[0].count, candidates_count = plan.symmetric ? queries_count : plan.side[1].count;
    if (!queries_count || !candidates_count || queries_count > SZS_FUSED_MOST_STRINGS_TWO_PASSES || candidates_count > SZS_FUSED_MOST_STRINGS_TWO_PASSES ||
        !plan.sequence)
        return (int)hipErrorInvalidValue;
    u32 const candidate_blocks = (candidates_count + SZS_CANDIDATES_PER_WORKGROUP - 1) / SZS_CANDIDATES_PER_WORKGROUP;
    // candidate blocks per workgroup: the rule of launch_myers_short (tens of thousands of short-lived workgroups share a query's table)
    u32 blocks_per_group = 1;
    int const pinned = szs_tuning_get(szs_knob_merge_k);
    if (pinned > 0) blocks_per_group = (u32)pinned < candidate_blocks ? (u32)pinned : candidate_blocks;
    else
        while (blocks_per_group < 8 && (u64)queries_count * ((candidate_blocks + 2 * blocks_per_group - 1) / (2 * blocks_per_group)) >= 16384)
            blocks_per_group *= 2;
    u32 const groups = (candidate_blocks + blocks_per_group - 1) / blocks_per_group;
    if ((u64)queries_count * groups > (1ull << 30)) return (int)hipErrorInvalidValue;
    plan.blocks_per_group = blocks_per_group;
    if (blocks_per_group > 1)
        hipLaunchKernelGGL(levenshtein_myers_short_fused_kernel<true>, dim3(queries_count * groups), dim3(256), 0, static_cast<hipStream_t>(stream), plan,
                           groups, results, results_row_stride, layout);
    else
        hipLaunchKernelGGL(levenshtein_myers_short_fused_kernel<false>, dim3(queries_count * groups), dim3(256), 0, static_cast<hipStream_t>(stream), plan,
                           groups, results, results_row_stride, layout);
    return (int)hipGetLastError();
}

extern "C" int szs_hip_levenshtein_myers_split(unsigned words, unsigned lanes, szs_string_ref_t const *queries, uint32_t queries_count,
                                               szs_string_ref_t const *candidates, uint32_t candidates_count, uint64_t *results,
                                               uint64_t results_row_stride, int symmetric, szs_ref_guard_t const *guard, void *stream) {
    using namespace szs_hip;
    if (!queries_count || !candidates_count) return 0;
    hipStream_t const s = static_cast<hipStream_t>(stream);
#define SZS_SPLIT_CASE(W, L)                                                                                           \
    if (words == W && lanes == L)                                                                                      \
        return launch_myers_split<W / L, L>(queries, queries_count, candidates, candidates_count, results, results_row_stride, symmetric, guard, s);
    SZS_SPLIT_CASE(16, 2) /* 16 words: only in launches of a few workgroups (host/plan.c: szs_plan_myers_shape) */
    SZS_SPLIT_CASE(16, 4)
    SZS_SPLIT_CASE(24, 2)
    SZS_SPLIT_CASE(32, 2)
    SZS_SPLIT_CASE(48, 2)
    SZS_SPLIT_CASE(64, 2)
    SZS_SPLIT_CASE(32, 4)
    SZS_SPLIT_CASE(48, 4)
    SZS_SPLIT_CASE(64, 4)
    SZS_SPLIT_CASE(32, 8)
    SZS_SPLIT_CASE(64, 8)
#undef SZS_SPLIT_CASE
    return (int)hipErrorInvalidValue;
}

extern "C" int szs_hip_levenshtein_myers_runes(szs_string_ref_t const *queries, uint32_t queries_count,
                                               szs_string_ref_t const *candidates, uint32_t candidates_count,
                                               uint64_t *results, uint64_t results_row_stride, int symmetric,
                                               uint32_t alphabet, void *stream) {
    using namespace szs_hip;
    if (!queries_count || !candidates_count) return 0;
    if (alphabet > SZS_ALPHABET_MOST) return (int)hipErrorInvalidValue;
    return launch_myers_short<true>(queries, queries_count, candidates, candidates_count, results, results_row_stride, symmetric, nullptr,
                                    static_cast<hipStream_t>(stream), alphabet ? ((size_t)alphabet + 1) * sizeof(u32) : 0, (u32)alphabet);
}

extern "C" size_t szs_hip_levenshtein_myers_banded_bytes(uint32_t queries_count, uint32_t candidates_count, uint32_t longest_candidate) {
    using namespace szs_hip;
    u64 const work_items = (u64)queries_count * ((candidates_count + SZS_CANDIDATES_PER_WORKGROUP - 1) / SZS_CANDIDATES_PER_WORKGROUP);
    return banded_header_bytes_k + (size_t)banded_grid(work_items) * (longest_candidate / 16 + 2) * 256 * sizeof(u32);
}

extern "C" int szs_hip_levenshtein_myers_banded(szs_string_ref_t const *queries, uint32_t queries_count,
                                                szs_string_ref_t const *candidates, uint32_t candidates_count,
                                                uint32_t longest_candidate, uint64_t *results, uint64_t results_row_stride,
                                                int symmetric, void *workspace, void *stream) {
    using namespace szs_hip;
    if (!queries_count || !candidates_count) return 0;
    u32 const candidate_blocks = (candidates_count + SZS_CANDIDATES_PER_WORKGROUP - 1) / SZS_CANDIDATES_PER_WORKGROUP;
    u64 const work_items = (u64)queries_count * candidate_blocks;
    if (work_items > 0xFFFFFFF0ull) return (int)hipErrorInvalidValue;
    hipStream_t const s = static_cast<hipStream_t>(stream);
    u32 *const counter = static_cast<u32 *>(workspace);
    hipError_t const error = hipMemsetAsync(counter, 0, sizeof(u32), s);
    if (error != hipSuccess) return (int)error;
    hipLaunchKernelGGL(levenshtein_myers_banded_kernel, dim3(banded_grid(work_items)), dim3(256), 0, s, queries, queries_count,
                       candidates, candidates_count, candidate_blocks, results, results_row_stride, symmetric,
                       reinterpret_cast<u32 *>(static_cast<char *>(workspace) + banded_header_bytes_k), longest_candidate / 16 + 2,
                       counter);
    return (int)hipGetLastError();
}

extern "C" size_t szs_hip_levenshtein_myers_banded_runes_bytes(uint32_t queries_count, uint32_t candidates_count,
                                                               uint32_t longest_candidate) {
    using namespace szs_hip;
    u32 rune_slots = 0, id_capacity = 0;
    size_t lds_bytes = 0;
    if (!rune_lds_plan_dense(banded_runes_words_k, rune_slots, id_capacity, lds_bytes)) return 0;
    u64 const work_items = (u64)queries_count * ((candidates_count + SZS_CANDIDATES_PER_WORKGROUP - 1) / SZS_CANDIDATES_PER_WORKGROUP);
    return banded_header_bytes_k + (size_t)banded_runes_grid(work_items, lds_bytes) * (longest_candidate / 16 + 2) * 256 * sizeof(u32);
}

extern "C" int szs_hip_levenshtein_myers_banded_runes(szs_string_ref_t const *queries, uint32_t queries_count,
                                                      szs_string_ref_t const *candidates, uint32_t candidates_count,
                                                      uint32_t longest_candidate, uint64_t *results, uint64_t results_row_stride,
                                                      int symmetric, void *workspace, void *stream) {
    using namespace szs_hip;
    if (!queries_count || !candidates_count) return 0;
    u32 rune_slots = 0, id_capacity = 0;
    size_t lds_bytes = 0;
    if (!rune_lds_plan_dense(banded_runes_words_k, rune_slots, id_capacity, lds_bytes)) return (int)hipErrorNotSupported;
    static int granted_on[device_slots_k];
    int *const granted = &granted_on[device_slot()];
    if (!cached(granted)) {
        hipError_t const error = hipFuncSetAttribute(reinterpret_cast<void const *>(levenshtein_myers_banded_runes_kernel<banded_runes_words_k>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)(((size_t)160 << 10) - 1024));
        if (error != hipSuccess) {
            (void)hipGetLastError();
            return (int)hipErrorNotSupported; // the host falls back to the rune-keyed DP kernel
        }
        remember(granted, 1);
    }
    u32 const candidate_blocks = (candidates_count + SZS_CANDIDATES_PER_WORKGROUP - 1) / SZS_CANDIDATES_PER_WORKGROUP;
    u64 const work_items = (u64)queries_count * candidate_blocks;
    if (work_items > 0xFFFFFFF0ull) return (int)hipErrorInvalidValue;
    hipStream_t const s = static_cast<hipStream_t>(stream);
    u32 *const counter = static_cast<u32 *>(workspace);
    hipError_t const error = hipMemsetAsync(counter, 0, sizeof(u32), s);
    if (error != hipSuccess) return (int)error;
    hipLaunchKernelGGL(levenshtein_myers_banded_runes_kernel<banded_runes_words_k>, dim3(banded_runes_grid(work_items, lds_bytes)), dim3(256),
                       lds_bytes, s, queries, queries_count, candidates, candidates_count, candidate_blocks, results, results_row_stride,
                       symmetric, reinterpret_cast<u32 *>(static_cast<char *>(workspace) + banded_header_bytes_k),
                       longest_candidate / 16 + 2, counter, rune_slots, id_capacity);
    return (int)hipGetLastError();
}

extern "C" int szs_hip_levenshtein_myers_runes_split(unsigned words, unsigned lanes, szs_string_ref_t const *queries,
                                                     uint32_t queries_count, szs_string_ref_t const *candidates,
                                                     uint32_t candidates_count, uint64_t *results, uint64_t results_row_stride,
                                                     int symmetric, uint32_t alphabet, uint32_t pairs_per_workgroup, void *stream) {
    using namespace szs_hip;
    if (!queries_count || !candidates_count) return 0;
    if (alphabet > SZS_ALPHABET_MOST) return (int)hipErrorInvalidValue;
    hipStream_t const s = static_cast<hipStream_t>(stream);
#define SZS_SPLIT_RUNES_CASE(W, L)                                                                                     \
    if (words == W && lanes == L)                                                                                      \
        return launch_split_runes<W / L, L>(queries, queries_count, candidates, candidates_count, results, results_row_stride, symmetric, alphabet, \
                                            pairs_per_workgroup, s);
    SZS_SPLIT_RUNES_CASE(16, 2)
    SZS_SPLIT_RUNES_CASE(16, 4)
    SZS_SPLIT_RUNES_CASE(24, 2)
    SZS_SPLIT_RUNES_CASE(32, 2)
    SZS_SPLIT_RUNES_CASE(48, 2)
    SZS_SPLIT_RUNES_CASE(64, 2)
    SZS_SPLIT_RUNES_CASE(32, 4)
    SZS_SPLIT_RUNES_CASE(48, 4)
    SZS_SPLIT_RUNES_CASE(64, 4)
#undef SZS_SPLIT_RUNES_CASE
    return (int)hipErrorInvalidValue;
}

extern "C" int szs_hip_levenshtein_myers_runes_long(unsigned words, szs_string_ref_t const *queries, uint32_t queries_count,
                                                    szs_string_ref_t const *candidates, uint32_t candidates_count,
                                                    uint64_t *results, uint64_t results_row_stride, int symmetric,
                                                    uint32_t alphabet, void *stream) {
    using namespace szs_hip;
    if (!queries_count || !candidates_count) return 0;
    if (alphabet > SZS_ALPHABET_MOST) return (int)hipErrorInvalidValue;
    hipStream_t const s = static_cast<hipStream_t>(stream);
#define SZS_MYERS_RUNES_CASE(W)                                                                                        \
    case W: return launch_long_runes<W>(queries, queries_count, candidates, candidates_count, results, results_row_stride, symmetric, alphabet, s);
    switch (words) {
        SZS_MYERS_RUNES_CASE(10)
        SZS_MYERS_RUNES_CASE(12)
        SZS_MYERS_RUNES_CASE(16)
        SZS_MYERS_RUNES_CASE(20)
        SZS_MYERS_RUNES_CASE(24)
        SZS_MYERS_RUNES_CASE(32)
        SZS_MYERS_RUNES_CASE(48)
        SZS_MYERS_RUNES_CASE(64)
    default: return (int)hipErrorInvalidValue;
    }
#undef SZS_MYERS_RUNES_CASE
}

/** The launch variant for a query of `words` 32-bit words: SZS_MYERS_SHORT_WORDS for everything the mixed-width kernel
 *  takes, else the next instantiated long width; 0 = too long for the bit-parallel kernels. */
extern "C" unsigned szs_hip_levenshtein_myers_round_words(unsigned words) {
    static unsigned const steps[] = {SZS_MYERS_SHORT_WORDS, 10, 12, 16, 20, 24, 32, 48, 64};
    for (unsigned i = 0; i < sizeof(steps) / sizeof(steps[0]); ++i)
        if (words <= steps[i]) return steps[i];
    return 0;
}
