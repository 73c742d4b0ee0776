/*
 *  myers_queue.hip - unit-cost byte-level Levenshtein distances of a MIXED-LENGTH batch in ONE persistent launch (round 4).
 *
 *  The reference runs every size tier of a call behind one trampoline with a single trailing synchronisation and cuts rows
 *  into (query, segment) work units so that the device stays full
 *      /root/reference/include/stringzillas/similarities/cuda.cuh:4435-4741 (trampoline), :2580-2586 (work units).
 *  Rounds 1-3 of this build launched one kernel per bit-vector width (hip/lev_myers.hip: up to nine launches over eight
 *  streams for a Zipf-length batch) and every launch ended in a tail of its own - an eighth of config 5, one GPU's share of
 *  eight, took 1.85 ms where 1.2 ms was the work (profiles/r03/shard_preview.jsonl).  Here:
 *
 *  - ONE launch, one persistent grid (two workgroups of 512 threads per CU: four wavefronts per SIMD, 64 KB of match masks
 *    each), one queue of work items in device memory: a ticket counter the workgroups draw from.
 *  - A work ITEM is (one query, S consecutive candidates of the length-sorted candidate array).  The workgroup builds the
 *    query's match masks once, then its eight WAVEFRONTS draw blocks of candidates from a counter in LDS - longest block
 *    first - so that a wavefront whose texts ended early takes the next block instead of waiting at a barrier for the
 *    wavefront with the longest texts (sorted Zipf lengths spread 2x over 512 neighbours).
 *  - The ORDER of the queue is planned on the host (host/plan.c: szs_plan_queue) from what the planner reports about the two
 *    sides - strings per width class, and the lengths at 33 ranks of each side: the (query slice) x (candidate column) TILES
 *    of the cross-product are sorted by the time one of their items holds a workgroup, longest first (longest-processing-time
 *    list scheduling), and handed to the kernel as an argument - 96 tiles x 28 bytes, no upload, no table in LDS.
 *  - The WIDTH is a per-item scalar decision, as in the short kernel of lev_myers.hip: up to 16 words a pair is one lane
 *    (bodies of exactly 1 ... 8, 10, 12, 16 words); wider patterns - or, in a call so short that one pair's columns would
 *    be a large part of it, any pattern - are spread over L = 2 ... 16 adjacent lanes of 4, 8, 12 or 16 words each, the
 *    strip pipeline of levenshtein_myers_split_kernel with L a run-time value (any L: 5 lanes x 4 words take a 20-word
 *    pattern exactly).  Every body stays under 128 registers, so all of them share the four wavefronts per SIMD.
 *
 *  Results are bit-identical to every other unit-cost kernel of this library and to the reference's serial scorer
 *  (levenshtein_distance_myers<char, serial>, serial.hpp:2073-2314): the same column update (hip/myers_core.hpp), the same
 *  right-aligned pattern over phantom rows, the same telescoped distance.
 */
#include "myers_core.hpp"

namespace szs_hip {

/** The workgroup's match-mask table: 64 words x 256 rows x 4 bytes of DYNAMIC LDS, so that the bodies below - functions of
 *  their own, see queue_lanes - address it as LDS (`ds_read_b128`) without being handed a generic pointer. */
extern __shared__ __attribute__((aligned(16))) u32 queue_arena[];

__device__ __forceinline__ u32 uniform(u32 value) { return __builtin_amdgcn_readfirstlane(value); }
__device__ __forceinline__ u64 uniform(u64 value) { return ((u64)uniform((u32)(value >> 32)) << 32) | uniform((u32)value); }
template <typename pointee_> __device__ __forceinline__ pointee_ *uniform(pointee_ *pointer) {
    return reinterpret_cast<pointee_ *>(uniform((u64) reinterpret_cast<uintptr_t>(pointer)));
}

constexpr u32 queue_threads_k = 512;        // eight wavefronts share one query's match masks
constexpr int queue_widest_k = 64;          // words of the widest pattern: 2048 bytes, 64 KB of masks
constexpr size_t queue_arena_bytes_k = (size_t)queue_widest_k * byte_rows_k * sizeof(u32);
constexpr u32 queue_pattern_reads_k = (32u * queue_widest_k + queue_threads_k - 1) / queue_threads_k; // pattern bytes per thread

/** dword index of word `w` of byte `row`'s mask: rows of `row_words` = 1, 2 or 4 words, chunk-major (peq_layout). */
__device__ __forceinline__ u32 queue_mask_index(u32 row_words, u32 row, u32 w) {
    return row_words == 4 ? (((w >> 2) * (u32)byte_rows_k + row) << 2) + (w & 3u) : row * row_words + w;
}

/**
 *  One wavefront, one candidate per lane: candidates [lo, hi) of the sorted array (at most 64) against the query whose masks
 *  are in `peq`.  The lane loop of lev_myers.hip's myers_workgroup: unpredicated batches while every live lane still has a
 *  whole batch of columns, then a tail predicated on each lane's own length. */
template <int words_, int text_dwords_>
__device__ __attribute__((noinline)) void queue_lanes(u32 table_offset, szs_string_ref_t query, szs_string_ref_t const *__restrict__ candidates,
                                                      u32 lo, u32 hi, u64 *__restrict__ results, u64 results_row_stride, int layout) {
    // A FUNCTION, not inlined: inside the persistent kernel the fifteen bodies shared one register allocation, and what the widest
    // of them needed spilled in the loops of the narrowest (1024 x 1024 x 128 bytes: 77 -> 56 TCUPS when the team bodies grew).
    // Arguments arrive in vector registers; what is uniform goes back to scalars here.
    u32 const *const peq = queue_arena + uniform(table_offset);
    query.address = uniform(query.address), query.length = uniform(query.length), query.index = uniform(query.index);
    candidates = uniform(candidates), results = uniform(results), results_row_stride = uniform(results_row_stride);
    lo = uniform(lo), hi = uniform(hi), layout = (int)uniform((u32)layout);
    u32 const slot = lo + (threadIdx.x & 63u);
    bool live = slot < hi;
    szs_string_ref_t candidate = {0, 0, 0};
    if (live) candidate = candidates[slot];
    if ((layout & SZS_LAYOUT_SYMMETRIC) && candidate.index > query.index) live = false; // upper triangle: mirrored from below
    u32 const query_length = query.length;
    u32 const pad = 32u * words_ - query_length; // phantom low rows
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
    auto take = [&](u32 symbol) {
        u32 eq[words_];
        load_match_masks<words_, byte_rows_k>(peq, symbol, eq);
        myers_column<words_>(vp, vn, eq);
    };

    // Lanes without a text stream from the query instead (always-valid memory; their symbols are never consumed)
    u64 const safe_address = text_length ? candidate.address : query.address;
    text_stream_t text(safe_address, text_length);
    if (!text_length) text.valid_dwords = query_length ? 1 : 0;
    u32 raw_low = text.raw(0);
    u32 column = 0, dword = 0;
    constexpr u32 columns_per_iteration = 4 * text_dwords_;
    constexpr bool clamped_reads = text_dwords_ == 1; // as measured for the one-dword loops of the long kernels
    if (columns_per_iteration <= shortest_in_wave && longest_in_wave && query_length) {
        u32 ahead[text_dwords_];
#pragma unroll
        for (int d = 0; d < text_dwords_; ++d) ahead[d] = clamped_reads ? text.raw_clamped(1 + d) : text.raw(1 + d);
        for (; column + columns_per_iteration <= shortest_in_wave; column += columns_per_iteration, dword += text_dwords_) {
            u32 symbols[text_dwords_];
            symbols[0] = text.splice(raw_low, ahead[0]);
#pragma unroll
            for (int d = 1; d < text_dwords_; ++d) symbols[d] = text.splice(ahead[d - 1], ahead[d]);
            raw_low = ahead[text_dwords_ - 1];
#pragma unroll
            for (int d = 0; d < text_dwords_; ++d)
                ahead[d] = clamped_reads ? text.raw_clamped(dword + text_dwords_ + 1 + d) : text.raw(dword + text_dwords_ + 1 + d);
            if constexpr (words_ <= 8) {
#pragma unroll
                for (int step = 0; step < 4 * text_dwords_; ++step) take((symbols[step / 4] >> (8 * (step % 4))) & 0xFFu);
            }
            else {
                // Ten words and more: left alone, hipcc hoists the mask reads of all four columns and the 16-word body spills 29 of
                // the 128 registers that four wavefronts per SIMD leave a lane; fenced column by column the reads wait for nothing
                // but also overlap nothing.  So only the FIRST chunk of the next column is fetched ahead (the carry chain starts
                // there); the other chunks are issued when the column begins and arrive under its first words.
                static_assert(text_dwords_ == 1, "four columns per iteration");
                uint4 const *const rows = reinterpret_cast<uint4 const *>(peq);
                constexpr int chunks = peq_layout<words_>::chunks;
                uint4 first = rows[symbols[0] & 0xFFu];
#pragma unroll
                for (int step = 0; step < 4; ++step) {
                    u32 const symbol = (symbols[0] >> (8 * step)) & 0xFFu;
                    u32 eq[words_];
                    eq[0] = first.x, eq[1] = first.y, eq[2] = first.z, eq[3] = first.w;
#pragma unroll
                    for (int chunk = 1; chunk < chunks; ++chunk) {
                        uint4 const row = rows[chunk * byte_rows_k + symbol];
                        if (chunk * 4 + 0 < words_) eq[chunk * 4 + 0] = row.x;
                        if (chunk * 4 + 1 < words_) eq[chunk * 4 + 1] = row.y;
                        if (chunk * 4 + 2 < words_) eq[chunk * 4 + 2] = row.z;
                        if (chunk * 4 + 3 < words_) eq[chunk * 4 + 3] = row.w;
                    }
                    if (step < 3) first = rows[(symbols[0] >> (8 * (step + 1))) & 0xFFu];
                    myers_column<words_>(vp, vn, eq);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    }
    if (column < longest_in_wave) { // ragged tail: one dword per iteration, each column predicated on this lane's own length
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

    if (live) {
        u32 distance = text_length;
#pragma unroll
        for (int w = 0; w < words_; ++w) distance += (u32)__builtin_popcount(vp[w]) - (u32)__builtin_popcount(vn[w]);
        bool const transposed = (layout & SZS_LAYOUT_TRANSPOSED) != 0; // kernel roles swapped by the host
        u64 const row = transposed ? candidate.index : query.index, column_of = transposed ? query.index : candidate.index;
        results[row * results_row_stride + column_of] = distance;
        if ((layout & SZS_LAYOUT_SYMMETRIC) && candidate.index != query.index) results[column_of * results_row_stride + row] = distance;
    }
}

/**
 *  The text of a lane that runs `delay` columns behind its team's first lane: the string as if it began `delay` bytes earlier.
 *  Only aligned dwords that hold at least one byte of the string are ever loaded (text_stream_t's rule: such a dword lies in a
 *  page the caller owns); the bytes "before" the string read as zero and belong to columns the lane does not score.
 */
struct delayed_text_t {
    u32 const *aligned_base;
    u32 byte_shift, first_valid, valid_dwords;
    __device__ __forceinline__ delayed_text_t(u64 address, u32 length, u32 delay) {
        u64 const from = address - delay;
        aligned_base = reinterpret_cast<u32 const *>(from & ~(u64)3);
        byte_shift = (u32)(from & 3);
        first_valid = (u32)(((address & ~(u64)3) - (from & ~(u64)3)) >> 2);
        valid_dwords = length ? (byte_shift + delay + length + 3) / 4 : 0;
    }
    __device__ __forceinline__ u32 raw(u32 dword_index) const {
        return dword_index >= first_valid && dword_index < valid_dwords ? aligned_base[dword_index] : 0u;
    }
    __device__ __forceinline__ u32 splice(u32 raw_low, u32 raw_high) const { return __builtin_amdgcn_alignbyte(raw_high, raw_low, byte_shift); }
};

/**
 *  One wavefront, `lanes` adjacent lanes per pair (2 ... 16, any value: the wavefront holds 64 / lanes teams, whatever is left
 *  over idles - the deltas move by `wave_shr:1`, which crosses the rows of 16 that `row_shr` stops at, so a team of 12 lanes
 *  wastes 4 lanes of 64, not 4 of 16): lane k of a team holds words [k w, (k + 1) w) of the pattern's bit-vector and runs k
 *  columns behind lane k - 1 - a systolic strip pipeline inside the wavefront, as in lev_myers.hip's
 *  levenshtein_myers_split_kernel.  What differs:
 *   - every lane reads the text itself, `k` bytes behind (delayed_text_t; the team's lanes hit the same cache lines), so only
 *     the two delta bits under a strip's last row travel - one `v_mov_b32_dpp row_shr:1` of a 2-bit value per column, no symbol,
 *     no valid flag, nothing to pack or unpack;
 *   - while EVERY lane of the wavefront is inside its text (from step lanes - 1 to the shortest text) four columns run without
 *     a predicate or a branch, their mask reads ahead of the arithmetic; only the fill and the drain test each column.
 *  With four words per lane the old form spent as many instructions on the hand-over as on the column (profiles/r04).
 *  Candidates [lo, hi): at most 64 / lanes of them.
 */
template <int words_per_lane_>
__device__ __attribute__((noinline)) void queue_team(u32 table_offset, u32 lanes, szs_string_ref_t query,
                                                     szs_string_ref_t const *__restrict__ candidates, u32 lo, u32 hi,
                                                     u64 *__restrict__ results, u64 results_row_stride, int layout) {
    static_assert(words_per_lane_ % 4 == 0, "whole 16-byte chunks of the masks per lane");
    u32 const *const peq = queue_arena + uniform(table_offset); // (a function of its own: see queue_lanes)
    query.address = uniform(query.address), query.length = uniform(query.length), query.index = uniform(query.index);
    candidates = uniform(candidates), results = uniform(results), results_row_stride = uniform(results_row_stride);
    lo = uniform(lo), hi = uniform(hi), layout = (int)uniform((u32)layout), lanes = uniform(lanes);
    constexpr int chunks_per_lane = words_per_lane_ / 4;
    u32 const lane = threadIdx.x & 63u;
    u32 const teams_per_wave = 64u / lanes;
    u32 const team = lane / lanes, part = lane - team * lanes;
    u32 const slot = lo + team;
    bool live = team < teams_per_wave && slot < hi;
    szs_string_ref_t candidate = {0, 0, 0};
    if (live) candidate = candidates[slot];
    if ((layout & SZS_LAYOUT_SYMMETRIC) && candidate.index > query.index) live = false;
    u32 const query_length = query.length;
    u32 const pad = 32u * words_per_lane_ * lanes - query_length; // phantom low rows: whole lanes of them are inert too
    u32 const text_length = live ? candidate.length : 0;
    u32 const longest_in_wave = wave_max_u32(text_length);
    u32 const shortest_in_wave = ~wave_max_u32(live ? ~text_length : 0u); // over live lanes; no live lane: ~0, unused
    bool const head = part == 0;

    u32 vp[words_per_lane_], vn[words_per_lane_];
#pragma unroll
    for (int w = 0; w < words_per_lane_; ++w) {
        u32 const first_bit = 32u * (part * words_per_lane_ + w);
        vp[w] = first_bit >= pad ? ~0u : (first_bit + 32u <= pad ? 0u : (~0u << (pad - first_bit)));
        vn[w] = 0;
    }
    // this lane's chunks of the masks: chunk c of byte s at uint4 index c * 256 + s
    uint4 const *const my_rows = reinterpret_cast<uint4 const *>(peq) + part * chunks_per_lane * byte_rows_k;
    auto masks_of = [&](u32 symbol, u32 (&eq)[words_per_lane_]) {
#pragma unroll
        for (int chunk = 0; chunk < chunks_per_lane; ++chunk) {
            uint4 const row = my_rows[chunk * byte_rows_k + symbol];
            eq[chunk * 4 + 0] = row.x, eq[chunk * 4 + 1] = row.y, eq[chunk * 4 + 2] = row.z, eq[chunk * 4 + 3] = row.w;
        }
    };

    delayed_text_t const text(candidate.address, text_length, part);
    u32 raw_low = text.raw(0), next = text.raw(1);
    u32 incoming = 0; // the two delta bits (hp | hn << 1) the lane below produced for the column this lane takes next
    u32 const steps = longest_in_wave ? longest_in_wave + lanes - 1 : 0;
#pragma unroll 1
    for (u32 step = 0, dword = 0; step < steps; step += 4, ++dword) {
        u32 const symbols = text.splice(raw_low, next);
        raw_low = next, next = text.raw(dword + 2);
        if (step + 1 >= lanes && step + 4 <= shortest_in_wave) { // every lane of every live team is inside its text: no tests
            auto column = [&](u32 const (&eq)[words_per_lane_]) {
                u32 const entering = head ? 1u : incoming; // DP row 0 grows by one per column
                u32 const leaving = myers_strip_column<words_per_lane_>(vp, vn, eq, entering & 1u, entering >> 1);
                // wave_shr:1 - every lane takes its lower neighbour's bits; lane 0 (always a head) takes zero
                incoming = (u32)__builtin_amdgcn_update_dpp(0, (int)leaving, 0x138, 0xF, 0xF, true);
            };
            if constexpr (words_per_lane_ <= 8) { // few enough registers: all four columns' masks ahead of the arithmetic
                u32 eq[4][words_per_lane_];
#pragma unroll
                for (u32 sub = 0; sub < 4; ++sub) masks_of((symbols >> (8 * sub)) & 0xFFu, eq[sub]);
#pragma unroll
                for (u32 sub = 0; sub < 4; ++sub) column(eq[sub]);
            }
            else { // twelve and sixteen words: only the FIRST chunk of the next column's masks is fetched ahead (the carry chain
                   // starts there; the other chunks arrive under it) - left alone, hipcc hoists the reads of all four columns
                   // and spills 41 / 86 registers; a whole column ahead still spills 26 / 81
                uint4 first = my_rows[symbols & 0xFFu];
#pragma unroll
                for (u32 sub = 0; sub < 4; ++sub) {
                    u32 const symbol = (symbols >> (8 * sub)) & 0xFFu;
                    u32 eq[words_per_lane_];
                    eq[0] = first.x, eq[1] = first.y, eq[2] = first.z, eq[3] = first.w;
#pragma unroll
                    for (int chunk = 1; chunk < chunks_per_lane; ++chunk) {
                        uint4 const row = my_rows[chunk * byte_rows_k + symbol];
                        eq[chunk * 4 + 0] = row.x, eq[chunk * 4 + 1] = row.y, eq[chunk * 4 + 2] = row.z, eq[chunk * 4 + 3] = row.w;
                    }
                    if (sub < 3) first = my_rows[(symbols >> (8 * (sub + 1))) & 0xFFu];
                    column(eq);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        else {
#pragma unroll
            for (u32 sub = 0; sub < 4; ++sub) {
                u32 leaving = 0;
                if (step + sub - part < text_length) { // this lane's column: step + sub - part (unsigned: not before its first)
                    u32 eq[words_per_lane_];
                    masks_of((symbols >> (8 * sub)) & 0xFFu, eq);
                    u32 const entering = head ? 1u : incoming;
                    leaving = myers_strip_column<words_per_lane_>(vp, vn, eq, entering & 1u, entering >> 1);
                }
                incoming = (u32)__builtin_amdgcn_update_dpp(0, (int)leaving, 0x138, 0xF, 0xF, true);
            }
        }
    }

    i32 const mine = [&] {
        i32 delta = 0;
#pragma unroll
        for (int w = 0; w < words_per_lane_; ++w) delta += (i32)__builtin_popcount(vp[w]) - (i32)__builtin_popcount(vn[w]);
        return delta;
    }();
    i32 delta = mine;
    for (u32 k = 1; k < lanes; ++k) delta += __shfl_down(mine, k, 64); // the head adds up its team (the lanes above it)
    if (live && head) {
        u64 const distance = (u64)((i64)text_length + delta);
        bool const transposed = (layout & SZS_LAYOUT_TRANSPOSED) != 0;
        u64 const row = transposed ? candidate.index : query.index, column_of = transposed ? query.index : candidate.index;
        results[row * results_row_stride + column_of] = distance;
        if ((layout & SZS_LAYOUT_SYMMETRIC) && candidate.index != query.index) results[column_of * results_row_stride + row] = distance;
    }
}

/** Words of the masks as the one-lane bodies see a pattern of `needed` words: its exact width up to 8, then 10, 12, 16, 20. */
__device__ __forceinline__ u32 queue_body_words(u32 needed) {
    return needed <= 8u ? needed : needed <= 10u ? 10u : needed <= 12u ? 12u : needed <= 16u ? 16u : 20u;
}

/**
 *  The persistent kernel.  `tickets` is a counter in device memory that is never reset: the host passes the value it holds when
 *  the launch begins (`ticket_base`; a launch takes exactly plan.items_total + gridDim.x tickets - every workgroup stops at
 *  the first ticket past the queue's end - so the host knows it without asking).
 *
 *  A work item is G queries x S candidates: the masks of all G queries are built side by side (64 / G words of the table
 *  each), and the eight wavefronts draw (candidate block, query) pairs - candidate blocks from the longest down, every query
 *  of the group against one block before the next.  A wavefront whose texts end early draws again instead of waiting: with
 *  one query per item a workgroup's wavefronts idled through two fifths of the top column of a Zipf batch.  A group whose
 *  queries do not fit the tile's shape (a plan made for another batch) is scored one query at a time, at a shape that takes it.
 */
__global__ __launch_bounds__(queue_threads_k, 4 /* wavefronts per SIMD: two workgroups per CU */) void levenshtein_myers_queue_kernel(
    szs_string_ref_t const *__restrict__ queries, szs_string_ref_t const *__restrict__ candidates, u64 *__restrict__ results,
    u64 results_row_stride, int layout, u32 *__restrict__ tickets, u32 ticket_base, u64 *__restrict__ trace, szs_queue_plan_t plan) {
    u32 *const peq = queue_arena; // peq_layout<queue_widest_k>::total_dwords dwords of dynamic LDS
    __shared__ u32 next_ticket, wave_ticket;

    u32 const tid = threadIdx.x;
    // measuring aid (`trace` knob): when every workgroup began and ended, in 100 MHz ticks, and how many items it took
    u64 const began = trace ? wall_clock64() : 0;
    u32 items_taken = 0;
    if (tid == 0) next_ticket = atomicAdd(tickets, 1u) - ticket_base;
    __syncthreads();
    u32 tile_index = 0;
    for (;;) {
        u32 const item = __builtin_amdgcn_readfirstlane(next_ticket);
        if (item >= plan.items_total) break;
        ++items_taken;
        // ---- ticket -> tile -> (queries [q_first, q_first + q_count), candidates [c_lo, c_hi)): tickets only grow, so the tile
        //      index only moves forward
        while (tile_index + 1 < plan.tiles_count && item >= plan.tiles[tile_index + 1].first_item) ++tile_index;
        szs_queue_tile_t const tile = plan.tiles[tile_index];
        u32 const per_group = tile.queries_per_item ? (tile.queries_per_item < 16u ? tile.queries_per_item : 16u) : 1u;
        u32 const groups = (tile.query_count + per_group - 1u) / per_group;
        u32 const local = item - tile.first_item;
        u32 const block = local / groups; // blocks of S candidates are cut from the column's END: heaviest first
        u32 const q_first = tile.query_first + (local - block * groups) * per_group;
        u32 const q_count = tile.query_first + tile.query_count - q_first < per_group ? tile.query_first + tile.query_count - q_first : per_group;
        u32 const c_hi = tile.candidate_end - block * tile.candidates_per_item;
        u32 const c_lo = c_hi - tile.candidate_first > tile.candidates_per_item ? c_hi - tile.candidates_per_item : tile.candidate_first;
        u32 tile_lanes = tile.lanes ? (tile.lanes < 16u ? tile.lanes : 16u) : 1u;
        u32 tile_words = tile.words_per_lane <= 4u ? 4u : tile.words_per_lane <= 8u ? 8u : tile.words_per_lane <= 12u ? 12u : 16u;

        u32 ahead = 0; // the next item's ticket: drawn while this one is scored, looked at afterwards - the round trip is hidden
        bool drawn_ahead = false, alone = false;
        u32 done = 0; // queries of the group that have been scored
        while (done < q_count) {
            // ---- this pass: the whole group side by side, or - when that did not fit - one query at a shape of its own
            u32 const together = alone ? 1u : q_count;
            u32 lanes = tile_lanes, words_per_lane = tile_words;
            u32 slot_words = ((u32)queue_widest_k / (alone ? 1u : per_group)) & ~3u; // words of the table each query of the pass gets
            if (alone) {
                u32 const needed = __builtin_amdgcn_readfirstlane((queries[q_first + done].length + 31u) / 32u);
                if (lanes == 1u && needed > 20u) lanes = (needed + 15u) / 16u, words_per_lane = 16u;
                if (lanes > 1u && words_per_lane * lanes < needed) words_per_lane = 16u, lanes = (needed + 15u) / 16u;
            }
            u32 const slot_dwords = slot_words * (u32)byte_rows_k, slot_bytes = slot_words * 32u;

            // ---- the match masks of the pass: every thread's pattern bytes are in flight while the table is cleared
            u32 mine[queue_pattern_reads_k], position_of[queue_pattern_reads_k], table_of[queue_pattern_reads_k];
            bool misfit = false;
#pragma unroll
            for (u32 k = 0; k < queue_pattern_reads_k; ++k) {
                u32 const v = tid + k * queue_threads_k, g = v / slot_bytes, i = v - g * slot_bytes;
                position_of[k] = ~0u;
                if (g >= together) continue;
                szs_string_ref_t const query = queries[q_first + done + g];
                u32 const needed = query.length ? (query.length + 31u) / 32u : 1u;
                u32 const body_words = lanes > 1u ? words_per_lane * lanes : queue_body_words(needed);
                if (body_words > slot_words || needed > body_words) misfit = true;
                else if (i < query.length) {
                    mine[k] = reinterpret_cast<u8 const *>(query.address)[i];
                    position_of[k] = 32u * body_words - query.length + i; // right-aligned over phantom low rows
                    table_of[k] = g * slot_dwords | (body_words >= 3u ? 4u : body_words) << 28;
                }
            }
            u32 const clear_dwords = together * slot_dwords;
            for (u32 i = tid * 4u; i < clear_dwords; i += queue_threads_k * 4u) *reinterpret_cast<uint4 *>(peq + i) = make_uint4(0, 0, 0, 0);
            bool const unfit = __syncthreads_or(misfit) != 0; // A: the table is clear, everybody has read `next_ticket`
            if (unfit && !alone) { // a group that does not fit side by side is scored query by query instead
                alone = true;
                continue;
            }
            if (tid == 0) {
                if (!drawn_ahead) ahead = atomicAdd(tickets, 1u);
                wave_ticket = 0;
            }
            drawn_ahead = true;
            if (!unfit) { // (a query no shape takes - beyond 2048 bytes - is not this kernel's: the host never queues one; it is left alone)
#pragma unroll
            for (u32 k = 0; k < queue_pattern_reads_k; ++k)
                if (position_of[k] != ~0u)
                    atomicOr(&peq[(table_of[k] & 0x0FFFFFFFu) + queue_mask_index(table_of[k] >> 28, mine[k], position_of[k] >> 5)],
                             1u << (position_of[k] & 31u));
            __syncthreads(); // B: the tables are complete

            // ---- the wavefronts draw (candidate block, query) pairs, longest block first, until the pass is through
            u32 const pairs_per_wave = 64u / lanes;
            u32 const wave_blocks = ((c_hi - c_lo + pairs_per_wave - 1u) / pairs_per_wave) * together;
            for (;;) {
                u32 drawn = 0;
                if ((tid & 63u) == 0) drawn = atomicAdd(&wave_ticket, 1u);
                drawn = __builtin_amdgcn_readfirstlane(drawn);
                if (drawn >= wave_blocks) break;
                u32 const candidate_block = drawn / together, g = drawn - candidate_block * together;
                u32 const hi = c_hi - candidate_block * pairs_per_wave;
                u32 const lo = hi - c_lo > pairs_per_wave ? hi - pairs_per_wave : c_lo;
                szs_string_ref_t const query = queries[q_first + done + g];
                u32 const table = g * slot_dwords;
#define SZS_QUEUE_LANES(W, D) queue_lanes<W, D>(table, query, candidates, lo, hi, results, results_row_stride, layout)
#define SZS_QUEUE_TEAM(W) queue_team<W>(table, lanes, query, candidates, lo, hi, results, results_row_stride, layout)
                if (lanes > 1u) {
                    switch (words_per_lane) {
                    case 4: SZS_QUEUE_TEAM(4); break;
                    case 8: SZS_QUEUE_TEAM(8); break;
                    case 12: SZS_QUEUE_TEAM(12); break;
                    default: SZS_QUEUE_TEAM(16); break;
                    }
                }
                else {
                    switch (queue_body_words(__builtin_amdgcn_readfirstlane(query.length ? (query.length + 31u) / 32u : 1u))) {
                    case 1: SZS_QUEUE_LANES(1, 2); break;
                    case 2: SZS_QUEUE_LANES(2, 2); break;
                    case 3: SZS_QUEUE_LANES(3, 2); break;
                    case 4: SZS_QUEUE_LANES(4, 2); break;
                    case 5: SZS_QUEUE_LANES(5, 2); break;
                    case 6: SZS_QUEUE_LANES(6, 2); break;
                    case 7: SZS_QUEUE_LANES(7, 2); break;
                    case 8: SZS_QUEUE_LANES(8, 2); break;
                    case 10: SZS_QUEUE_LANES(10, 1); break;
                    case 12: SZS_QUEUE_LANES(12, 1); break;
                    case 16: SZS_QUEUE_LANES(16, 1); break;
                    default: SZS_QUEUE_LANES(20, 1); break;
                    }
                }
#undef SZS_QUEUE_LANES
#undef SZS_QUEUE_TEAM
            }
            } // fit
            done += together;
            if (done >= q_count && tid == 0) next_ticket = ahead - ticket_base;
            __syncthreads(); // C: nobody reads the tables any more; after the last pass the next ticket is visible
        }
    }
    if (trace && tid == 0) trace[3 * blockIdx.x] = began, trace[3 * blockIdx.x + 1] = wall_clock64(), trace[3 * blockIdx.x + 2] = items_taken;
}

/** Workgroups the device keeps resident (two per CU with 64 KB of LDS each), per device ordinal. */
static u32 queue_grid(u64 items) {
    static int resident_of[device_slots_k];
    int *const slot = &resident_of[device_slot()];
    int resident = cached(slot);
    if (!resident) {
        int device = 0, units = 0, per_unit = 0;
        if (hipGetDevice(&device) != hipSuccess ||
            hipDeviceGetAttribute(&units, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<void const *>(levenshtein_myers_queue_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)queue_arena_bytes_k) != hipSuccess ||
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_unit, levenshtein_myers_queue_kernel, (int)queue_threads_k, queue_arena_bytes_k) != hipSuccess ||
            units <= 0 || per_unit <= 0) {
            (void)hipGetLastError();
            units = 256, per_unit = 2;
        }
        resident = units * per_unit;
        remember(slot, resident);
    }
    return (u32)(items < (u64)resident ? items : (u64)resident);
}

} // namespace szs_hip

extern "C" unsigned szs_hip_levenshtein_myers_queue_grid(uint64_t items) { return szs_hip::queue_grid(items); }

extern "C" int szs_hip_levenshtein_myers_queue(szs_queue_plan_t const *plan, szs_string_ref_t const *queries,
                                               szs_string_ref_t const *candidates, uint64_t *results, uint64_t results_row_stride,
                                               int layout, uint32_t *tickets, uint32_t ticket_base, uint32_t *tickets_taken,
                                               uint64_t *trace, void *stream) {
    using namespace szs_hip;
    *tickets_taken = 0;
    if (!plan->items_total) return 0;
    if (plan->tiles_count > SZS_QUEUE_MOST_TILES) return (int)hipErrorInvalidValue;
    u32 const grid = queue_grid(plan->items_total);
    hipLaunchKernelGGL(levenshtein_myers_queue_kernel, dim3(grid), dim3(queue_threads_k), queue_arena_bytes_k, static_cast<hipStream_t>(stream), queries,
                       candidates, results, results_row_stride, layout, tickets, ticket_base, trace, *plan);
    hipError_t const error = hipGetLastError();
    if (error == hipSuccess) *tickets_taken = plan->items_total + grid;
    return (int)error;
}
