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

/** dword index of word `w` of symbol `row`'s mask: rows of `row_words` = 1, 2 or 4 words, chunk-major (peq_layout), `rows` per chunk. */
__device__ __forceinline__ u32 queue_mask_index(u32 row_words, u32 rows, u32 row, u32 w) {
    return row_words == 4 ? (((w >> 2) * rows + row) << 2) + (w & 3u) : row * row_words + w;
}

/* ---- codepoints: what differs from bytes ------------------------------------------------------------------------------------
 *
 *  The strings are UTF-32 arrays whose runes a renumbering pass (hip/utf8.hip) has replaced by dense ids 1 ... A, so a symbol
 *  indexes a table directly.  A query's masks are
 *    DIRECT   rows of 4 words per (chunk, id): (A + 1) x words x 4 bytes - 59 KB for 16 words at A = 926 (config 5u), what one
 *             lane per pair takes; read like the byte tables, one `ds_read_b128` per chunk;
 *    SPARSE   when that does not fit - a 2048-rune query at A = 926 would be 237 KB - the table of lev_myers.hip's rune_masks_t
 *             without its first level: `pointers[id][lane of the team][chunk of the lane]`, 16 bits each - bit 15 "claimed", the
 *             rest a slot of the POOL of 16-byte chunks, slot 0 the all-zero chunk - then the pool: a pattern of n runes sets n
 *             bits, so at most n chunks are non-zero whatever the alphabet.  29.7 KB + 32.8 KB at 64 words.  Two dependent LDS
 *             reads per column where the kernels of rounds 1-3 had three (id, pointer row, chunk), the first of them - the
 *             pointers of all four columns of a batch - issued before the batch's arithmetic begins.
 *  A lane reads its text 16 bytes at a time; a lane that runs k columns behind its team's first reads from k runes earlier,
 *  element by element where that window would leave the array (its first and last batch).
 */
constexpr size_t queue_rune_arena_bytes_k = 72u << 10; // two workgroups per CU still fit (160 KB); sparse tables of 60 words need 68 KB

/** Bytes of pointers a symbol takes per lane of a team in a sparse table: the lane's chunks as 16-bit values, padded to one read. */
__host__ __device__ constexpr u32 queue_pointer_bytes(u32 words_per_lane) { return words_per_lane <= 4 ? 2u : words_per_lane <= 8 ? 4u : 8u; }

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

/** Four columns of a lane's text at a time, for bytes (spliced dwords) and for codepoints (16-byte loads), `delay` columns
 *  behind the string's own start. */
template <bool runes_>
struct queue_text_t;

template <>
struct queue_text_t<false> {
    delayed_text_t text;
    u32 raw_low, next;
    __device__ __forceinline__ queue_text_t(u64 address, u32 length, u32 delay) : text(address, length, delay) { raw_low = text.raw(0), next = text.raw(1); }
    /** symbols of this lane's columns [4 batch, 4 batch + 4), one byte each; call with batch = 0, 1, 2 ... in order */
    __device__ __forceinline__ u32 take(u32 batch) {
        u32 const symbols = text.splice(raw_low, next);
        raw_low = next, next = text.raw(batch + 2);
        return symbols;
    }
    __device__ static __forceinline__ u32 symbol(u32 symbols, u32 sub) { return (symbols >> (8 * sub)) & 0xFFu; }
    using batch_t = u32;
};

template <>
struct queue_text_t<true> {
    u32 const *runes; // the string's first rune
    u32 length, padded, delay;
    uint4 ahead;
    __device__ __forceinline__ uint4 load(u32 first) const { // this lane's columns [first, first + 4): runes [first - delay, ...)
        u32 const element = first - delay;                   // (unsigned: before the string's start it is huge)
        uint4 quad;
        if (first >= delay && element + 4u <= padded) { // one 16-byte load at a 4-byte aligned address
            __builtin_memcpy(&quad, runes + element, sizeof(quad));
            return quad;
        }
        quad.x = element < length ? runes[element] : 0u, quad.y = element + 1u < length ? runes[element + 1u] : 0u;
        quad.z = element + 2u < length ? runes[element + 2u] : 0u, quad.w = element + 3u < length ? runes[element + 3u] : 0u;
        return quad;
    }
    __device__ __forceinline__ queue_text_t(u64 address, u32 length_, u32 delay_)
        : runes(reinterpret_cast<u32 const *>(address)), length(length_), padded((length_ + 3u) & ~3u), delay(delay_) {
        ahead = load(0); // (every UTF-32 array owns its storage up to the next 16-byte boundary: hip/kernels.h)
    }
    __device__ __forceinline__ uint4 take(u32 batch) {
        uint4 const now = ahead;
        ahead = load(4u * (batch + 1u));
        return now;
    }
    __device__ static __forceinline__ u32 symbol(uint4 const &symbols, u32 sub) { return sub == 0 ? symbols.x : sub == 1 ? symbols.y : sub == 2 ? symbols.z : symbols.w; }
    using batch_t = uint4;
};

/**
 *  Where a lane finds the masks of a symbol.  `prepare` is issued for all four columns of a batch before their arithmetic (for
 *  the sparse tables it is the first of two dependent reads); `chunk` reads 16 bytes of masks.
 */
template <int words_, bool sparse_>
struct queue_masks_t {
    static constexpr int chunks = (words_ + 3) / 4;
    uint4 const *rows; // direct: this lane's first chunk of symbol 0; sparse: the pool
    u32 stride;        // direct: uint4s from one chunk of a symbol to the next chunk of the same symbol (= rows of the table)
    char const *pointers; // sparse: this lane's pointers of symbol 0
    u32 pointer_stride;   // sparse: bytes from one symbol's pointers to the next's
    struct handle_t { u32 low, high; };
    __device__ __forceinline__ handle_t prepare(u32 symbol) const {
        handle_t handle = {symbol, 0};
        if constexpr (sparse_) {
            char const *const at = pointers + symbol * pointer_stride;
            if constexpr (words_ <= 4) handle.low = *reinterpret_cast<uint16_t const *>(at);
            else if constexpr (words_ <= 8) handle.low = *reinterpret_cast<u32 const *>(at);
            else {
                uint2 const both = *reinterpret_cast<uint2 const *>(at);
                handle.low = both.x, handle.high = both.y;
            }
        }
        return handle;
    }
    __device__ __forceinline__ uint4 chunk(handle_t const &handle, int index) const {
        if constexpr (sparse_) {
            u32 const pair = index < 2 ? handle.low : handle.high;
            u32 const slot = (index & 1 ? pair >> 16 : pair) & 0x7FFFu; // bit 15: "claimed" while the table was built
            return rows[slot];
        }
        else return rows[index * stride + handle.low];
    }
    __device__ __forceinline__ void load(handle_t const &handle, u32 (&eq)[words_], int from = 0) const {
#pragma unroll
        for (int index = from; index < chunks; ++index) {
            uint4 const row = chunk(handle, index);
            if (index * 4 + 0 < words_) eq[index * 4 + 0] = row.x;
            if (index * 4 + 1 < words_) eq[index * 4 + 1] = row.y;
            if (index * 4 + 2 < words_) eq[index * 4 + 2] = row.z;
            if (index * 4 + 3 < words_) eq[index * 4 + 3] = row.w;
        }
    }
};

/** What a body is told about the table of its query (uniform values, packed by the kernel). */
struct queue_table_t {
    u32 offset;  // dwords from the arena's start to the query's table
    u32 rows;    // symbols + 1: 256 for bytes, the renumbered alphabet + 1 for codepoints
    u32 pool;    // sparse: dwords from the table's start to its pool
};

/**
 *  One wavefront, one candidate per lane: candidates [lo, hi) of the sorted array (at most 64) against the query whose masks
 *  are at `table`.  The lane loop of lev_myers.hip's myers_workgroup: unpredicated batches while every live lane still has a
 *  whole batch of columns, then a tail predicated on each lane's own length.
 *
 *  A FUNCTION, not inlined: inside the persistent kernel the bodies shared one register allocation, and what the widest of them
 *  needed spilled in the loops of the narrowest (1024 x 1024 x 128 bytes: 77 -> 56 TCUPS when the team bodies grew).
 *  Arguments arrive in vector registers; what is uniform goes back to scalars here.
 */
template <int words_, int text_dwords_, bool runes_>
__device__ __attribute__((noinline)) void queue_lanes(queue_table_t table, szs_string_ref_t query, szs_string_ref_t const *__restrict__ candidates,
                                                      u32 lo, u32 hi, u64 *__restrict__ results, u64 results_row_stride, int layout) {
    u32 const *const peq = queue_arena + uniform(table.offset);
    u32 const rows = runes_ ? uniform(table.rows) : (u32)byte_rows_k;
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
    constexpr int table_words = runes_ && words_ < 3 ? 4 : words_; // codepoint tables always have 16-byte rows
    auto take = [&](u32 symbol) {
        u32 eq[table_words];
        if constexpr (runes_) {
            uint4 const *const rows_of = reinterpret_cast<uint4 const *>(peq);
#pragma unroll
            for (int chunk = 0; chunk < (table_words + 3) / 4; ++chunk) {
                uint4 const row = rows_of[chunk * rows + symbol];
                if (chunk * 4 + 0 < table_words) eq[chunk * 4 + 0] = row.x;
                if (chunk * 4 + 1 < table_words) eq[chunk * 4 + 1] = row.y;
                if (chunk * 4 + 2 < table_words) eq[chunk * 4 + 2] = row.z;
                if (chunk * 4 + 3 < table_words) eq[chunk * 4 + 3] = row.w;
            }
        }
        else load_match_masks<words_, byte_rows_k>(peq, symbol, eq);
        u32 narrow[words_];
#pragma unroll
        for (int w = 0; w < words_; ++w) narrow[w] = eq[w];
        myers_column<words_>(vp, vn, narrow);
    };
    // Ten words and more: left alone, hipcc hoists the mask reads of all four columns and the 16-word body spills 29 of the 128
    // registers that four wavefronts per SIMD leave a lane; fenced column by column the reads wait for nothing but also overlap
    // nothing.  So only the FIRST chunk of the next column is fetched ahead (the carry chain starts there); the other chunks are
    // issued when the column begins and arrive under its first words.
    auto take_four_wide = [&](u32 s0, u32 s1, u32 s2, u32 s3) {
        if constexpr (words_ > 8) {
        uint4 const *const rows_of = reinterpret_cast<uint4 const *>(peq);
        constexpr int chunks = (words_ + 3) / 4;
        u32 const symbols[4] = {s0, s1, s2, s3};
        uint4 first = rows_of[s0];
#pragma unroll
        for (int step = 0; step < 4; ++step) {
            u32 eq[words_];
            eq[0] = first.x, eq[1] = first.y, eq[2] = first.z, eq[3] = first.w;
#pragma unroll
            for (int chunk = 1; chunk < chunks; ++chunk) {
                uint4 const row = rows_of[chunk * rows + symbols[step]];
                if (chunk * 4 + 0 < words_) eq[chunk * 4 + 0] = row.x;
                if (chunk * 4 + 1 < words_) eq[chunk * 4 + 1] = row.y;
                if (chunk * 4 + 2 < words_) eq[chunk * 4 + 2] = row.z;
                if (chunk * 4 + 3 < words_) eq[chunk * 4 + 3] = row.w;
            }
            if (step < 3) first = rows_of[symbols[step + 1]];
            myers_column<words_>(vp, vn, eq);
            __builtin_amdgcn_sched_barrier(0);
        }
        }
    };

    u32 column = 0;
    if constexpr (runes_) {
        // ---- codepoints: FOUR columns per 16-byte load, loaded one batch early (lev_myers.hip: myers_workgroup)
        u32 const *const runes = reinterpret_cast<u32 const *>(candidate.address);
        auto quad_at = [&](u32 index) -> uint4 { // a multiple of 4; symbols past the text's end are never consumed
            return index < text_length ? *reinterpret_cast<uint4 const *>(runes + index) : make_uint4(0, 0, 0, 0);
        };
        if (4 <= shortest_in_wave && longest_in_wave) {
            uint4 ahead = quad_at(0);
            for (; column + 4 <= shortest_in_wave; column += 4) {
                uint4 const now = ahead;
                ahead = quad_at(column + 4);
                if constexpr (words_ <= 8) take(now.x), take(now.y), take(now.z), take(now.w);
                else take_four_wide(now.x, now.y, now.z, now.w);
            }
        }
#pragma unroll 1
        for (; column < longest_in_wave; ++column)
            if (column < text_length) take(runes[column]);
    }
    else {
        // Lanes without a text stream from the query instead (always-valid memory; their symbols are never consumed)
        u64 const safe_address = text_length ? candidate.address : query.address;
        text_stream_t text(safe_address, text_length);
        if (!text_length) text.valid_dwords = query_length ? 1 : 0;
        u32 raw_low = text.raw(0);
        u32 dword = 0;
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
                    static_assert(words_ <= 8 || text_dwords_ == 1, "four columns per iteration");
                    take_four_wide(symbols[0] & 0xFFu, (symbols[0] >> 8) & 0xFFu, (symbols[0] >> 16) & 0xFFu, symbols[0] >> 24);
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
 *  One wavefront, `lanes` adjacent lanes per pair (2 ... 16, any value: the wavefront holds 64 / lanes teams, whatever is left
 *  over idles - the deltas move by `wave_shr:1`, which crosses the rows of 16 that `row_shr` stops at, so a team of 12 lanes
 *  wastes 4 lanes of 64, not 4 of 16): lane k of a team holds words [k w, (k + 1) w) of the pattern's bit-vector and runs k
 *  columns behind lane k - 1 - a systolic strip pipeline inside the wavefront, as in lev_myers.hip's
 *  levenshtein_myers_split_kernel.  What differs:
 *   - every lane reads the text itself, `k` symbols behind (the team's lanes hit the same cache lines), so only the two delta
 *     bits under a strip's last row travel - one `v_mov_b32_dpp wave_shr:1` of a 2-bit value per column, no symbol, no valid
 *     flag, nothing to pack or unpack;
 *   - while EVERY lane of the wavefront is inside its text (from step lanes - 1 to the shortest text) four columns run without
 *     a predicate or a branch, their mask reads ahead of the arithmetic; only the fill and the drain test each column.
 *  With four words per lane the old form spent as many instructions on the hand-over as on the column (profiles/r04).
 *  Candidates [lo, hi): at most 64 / lanes of them.  A function of its own: see queue_lanes.
 */
template <int words_per_lane_, bool runes_, bool sparse_>
__device__ __attribute__((noinline)) void queue_team(queue_table_t table, u32 lanes, szs_string_ref_t query,
                                                     szs_string_ref_t const *__restrict__ candidates, u32 lo, u32 hi,
                                                     u64 *__restrict__ results, u64 results_row_stride, int layout) {
    static_assert(words_per_lane_ % 4 == 0, "whole 16-byte chunks of the masks per lane");
    static_assert(runes_ || !sparse_, "byte tables are always direct");
    constexpr int chunks_per_lane = words_per_lane_ / 4;
    u32 const *const peq = queue_arena + uniform(table.offset);
    u32 const rows = runes_ ? uniform(table.rows) : (u32)byte_rows_k;
    query.address = uniform(query.address), query.length = uniform(query.length), query.index = uniform(query.index);
    candidates = uniform(candidates), results = uniform(results), results_row_stride = uniform(results_row_stride);
    lo = uniform(lo), hi = uniform(hi), layout = (int)uniform((u32)layout), lanes = uniform(lanes);
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
    using masks_t = queue_masks_t<words_per_lane_, sparse_>;
    masks_t masks;
    if constexpr (sparse_) {
        masks.rows = reinterpret_cast<uint4 const *>(peq + uniform(table.pool));
        masks.pointer_stride = lanes * queue_pointer_bytes(words_per_lane_);
        masks.pointers = reinterpret_cast<char const *>(peq) + part * queue_pointer_bytes(words_per_lane_);
        masks.stride = 0;
    }
    else { // this lane's chunks of the masks: chunk c of symbol s at uint4 index c * rows + s
        masks.rows = reinterpret_cast<uint4 const *>(peq) + part * chunks_per_lane * rows;
        masks.stride = rows, masks.pointers = nullptr, masks.pointer_stride = 0;
    }

    using text_t = queue_text_t<runes_>;
    text_t text(candidate.address, text_length, part);
    u32 incoming = 0; // the two delta bits (hp | hn << 1) the lane below produced for the column this lane takes next
    u32 const steps = longest_in_wave ? longest_in_wave + lanes - 1 : 0;
#pragma unroll 1
    for (u32 step = 0, batch = 0; step < steps; step += 4, ++batch) {
        typename text_t::batch_t const symbols = text.take(batch);
        typename masks_t::handle_t handles[4];
#pragma unroll
        for (u32 sub = 0; sub < 4; ++sub) handles[sub] = masks.prepare(text_t::symbol(symbols, sub));
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
                for (u32 sub = 0; sub < 4; ++sub) masks.load(handles[sub], eq[sub]);
#pragma unroll
                for (u32 sub = 0; sub < 4; ++sub) column(eq[sub]);
            }
            else { // twelve and sixteen words: only the FIRST chunk of the next column's masks is fetched ahead (the carry chain
                   // starts there; the other chunks arrive under it) - left alone, hipcc hoists the reads of all four columns
                   // and spills 41 / 86 registers; a whole column ahead still spills 26 / 81
                uint4 first = masks.chunk(handles[0], 0);
#pragma unroll
                for (u32 sub = 0; sub < 4; ++sub) {
                    u32 eq[words_per_lane_];
                    eq[0] = first.x, eq[1] = first.y, eq[2] = first.z, eq[3] = first.w;
                    masks.load(handles[sub], eq, 1);
                    if (sub < 3) first = masks.chunk(handles[sub + 1], 0);
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
                    masks.load(handles[sub], eq);
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

/** Words of the masks as the one-lane bodies see a pattern of `needed` words: its exact width up to 8, then 10, 12, 16, 20
 *  (codepoints: 12, 16 - and none beyond, their direct tables would not fit). */
__device__ __forceinline__ u32 queue_body_words(u32 needed, bool runes) {
    if (runes) return needed <= 8u ? needed : needed <= 12u ? 12u : 16u;
    return needed <= 8u ? needed : needed <= 10u ? 10u : needed <= 12u ? 12u : needed <= 16u ? 16u : 20u;
}

/**
 *  The persistent kernel.  `tickets` is a counter in device memory that is never reset: the host passes the value it holds when
 *  the launch begins (`ticket_base`; a launch takes exactly plan.items_total + gridDim.x tickets - every workgroup stops at
 *  the first ticket past the queue's end - so the host knows it without asking).
 *
 *  A work item is G queries x S candidates: the masks of all G queries are built side by side (an equal share of the table
 *  each), and the eight wavefronts draw (candidate block, query) pairs - candidate blocks from the longest down, every query
 *  of the group against one block before the next.  A wavefront whose texts end early draws again instead of waiting: with
 *  one query per item a workgroup's wavefronts idled through two fifths of the top column of a Zipf batch.  A group whose
 *  queries do not fit the tile's shape (a plan made for another batch) is scored one query at a time, at a shape that takes it;
 *  a codepoint query that no table of this kernel takes raises `*unfit` (pinned host memory) and the host scores the call with
 *  the per-width launches instead.
 *
 *  `runes_`: strings are UTF-32 arrays of ids 1 ... `alphabet` (hip/utf8.hip renumbered the batch); tables as described above.
 */
template <bool runes_>
__global__ __launch_bounds__(queue_threads_k, 4 /* wavefronts per SIMD: two workgroups per CU */) void levenshtein_myers_queue_kernel(
    szs_string_ref_t const *__restrict__ queries, szs_string_ref_t const *__restrict__ candidates, u64 *__restrict__ results,
    u64 results_row_stride, int layout, u32 *__restrict__ tickets, u32 ticket_base, u64 *__restrict__ trace, u32 alphabet,
    u32 *__restrict__ unfit_flag, u32 unfit_sequence, szs_queue_plan_t plan) {
    u32 *const peq = queue_arena;
    __shared__ u32 next_ticket, wave_ticket, pool_used[16];
    constexpr u32 arena_dwords = (u32)((runes_ ? queue_rune_arena_bytes_k : queue_arena_bytes_k) / 4);
    u32 const rows = runes_ ? alphabet + 1u : (u32)byte_rows_k;

    u32 const tid = threadIdx.x;
    // measuring aid (`trace` knob): when every workgroup began and ended, in 100 MHz ticks, and how many items it took
    u64 const began = trace ? wall_clock64() : 0;
    u32 items_taken = 0, last_item = 0, longest_item = 0;
    u64 last_began = 0, longest_took = 0;
    if (tid == 0) next_ticket = atomicAdd(tickets, 1u) - ticket_base;
    __syncthreads();
    u32 tile_index = 0;
    for (;;) {
        u32 const item = __builtin_amdgcn_readfirstlane(next_ticket);
        if (item >= plan.items_total) break;
        ++items_taken;
        if (trace) {
            u64 const now = wall_clock64();
            if (items_taken > 1 && now - last_began > longest_took) longest_took = now - last_began, longest_item = last_item;
            last_item = item, last_began = now;
        }
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
        u32 const tile_lanes = tile.lanes ? (tile.lanes < 16u ? tile.lanes : 16u) : 1u;
        u32 const tile_words = tile.words_per_lane <= 4u ? 4u : tile.words_per_lane <= 8u ? 8u : tile.words_per_lane <= 12u ? 12u : 16u;
        bool const tile_sparse = runes_ && (tile.flags & SZS_QUEUE_TILE_SPARSE) != 0;

        u32 ahead = 0; // the next item's ticket: drawn while this one is scored, looked at afterwards - the round trip is hidden
        bool drawn_ahead = false, alone = false;
        u32 done = 0; // queries of the group that have been scored
        while (done < q_count) {
            // ---- this pass: the whole group side by side, or - when that did not fit - one query at a shape of its own
            u32 const together = alone ? 1u : q_count;
            u32 lanes = tile_lanes, words_per_lane = tile_words;
            bool sparse = tile_sparse;
            u32 const share = alone ? 1u : per_group;
            u32 const slot_positions = (((u32)queue_widest_k / share) & ~3u) * 32u; // pattern symbols each query of the pass may hold
            u32 const slot_dwords = (arena_dwords / share) & ~3u;                    // dwords of the table each query of the pass gets
            if (alone) {
                u32 const needed = __builtin_amdgcn_readfirstlane((queries[q_first + done].length + 31u) / 32u);
                u32 const widest_alone = runes_ ? 16u : 20u;
                if (lanes == 1u && needed > widest_alone) lanes = (needed + 15u) / 16u, words_per_lane = 16u;
                if (lanes > 1u && words_per_lane * lanes < needed) words_per_lane = 16u, lanes = (needed + 15u) / 16u;
                if (runes_) { // the table a lone query gets: direct when it fits the whole arena, sparse otherwise (teams only)
                    u32 const table_words = lanes > 1u ? words_per_lane * lanes : (queue_body_words(needed, true) + 3u) & ~3u;
                    sparse = rows * table_words > arena_dwords;
                    if (sparse && lanes == 1u) lanes = 2u, words_per_lane = needed <= 8u ? 4u : needed <= 16u ? 8u : needed <= 24u ? 12u : 16u,
                                               lanes = (needed + words_per_lane - 1u) / words_per_lane < 2u ? 2u : (needed + words_per_lane - 1u) / words_per_lane;
                }
            }
            u32 const pointer_dwords = sparse ? (rows * lanes * queue_pointer_bytes(words_per_lane) + 15u) / 16u * 4u : 0u; // 16-byte aligned pool

            // ---- the match masks of the pass: every thread's pattern symbols are in flight while the table is cleared
            u32 mine[queue_pattern_reads_k], position_of[queue_pattern_reads_k], table_of[queue_pattern_reads_k];
            bool misfit = false;
#pragma unroll
            for (u32 k = 0; k < queue_pattern_reads_k; ++k) {
                u32 const v = tid + k * queue_threads_k, g = v / slot_positions, i = v - g * slot_positions;
                position_of[k] = ~0u;
                if (g >= together) continue;
                szs_string_ref_t const query = queries[q_first + done + g];
                u32 const needed = query.length ? (query.length + 31u) / 32u : 1u;
                u32 const body_words = lanes > 1u ? words_per_lane * lanes : queue_body_words(needed, runes_);
                u32 const row_words = runes_ || body_words >= 3u ? 4u : body_words;
                u32 const table_dwords = sparse ? pointer_dwords + (query.length + 1u) * 4u
                                                : row_words == 4u ? rows * ((body_words + 3u) & ~3u) : rows * row_words;
                if (needed > body_words || query.length > slot_positions || table_dwords > slot_dwords) misfit = true;
                else if (i < query.length) {
                    mine[k] = runes_ ? reinterpret_cast<u32 const *>(query.address)[i] : (u32) reinterpret_cast<u8 const *>(query.address)[i];
                    position_of[k] = 32u * body_words - query.length + i; // right-aligned over phantom low rows
                    table_of[k] = g * slot_dwords | row_words << 28;
                    if (runes_ && mine[k] >= rows) misfit = true; // an id beyond the alphabet the launch was shaped for: never indexed
                }
            }
            u32 const clear_dwords = together * slot_dwords;
            for (u32 i = tid * 4u; i < clear_dwords; i += queue_threads_k * 4u) *reinterpret_cast<uint4 *>(peq + i) = make_uint4(0, 0, 0, 0);
            if (runes_ && tid < 16u) pool_used[tid] = 0;
            bool const unfit = __syncthreads_or(misfit) != 0; // A: the table is clear, everybody has read `next_ticket`
            if (unfit && !alone) { // a group that does not fit side by side is scored query by query instead
                alone = true;
                continue;
            }
            if (tid == 0) {
                if (!drawn_ahead) ahead = atomicAdd(tickets, 1u);
                wave_ticket = 0;
                // a query no table takes: bytes beyond 2048 (the host never queues one); codepoints whose alphabet outgrew the
                // arena - the host sees the flag after the call's wait and scores the batch with the per-width launches
                if (unfit && unfit_flag) *unfit_flag = unfit_sequence;
            }
            drawn_ahead = true;
            if (!unfit) {
            if (runes_ && sparse) { // claim a pool chunk for every (symbol, chunk) the patterns touch, then fill the chunks
#pragma unroll
                for (u32 k = 0; k < queue_pattern_reads_k; ++k)
                    if (position_of[k] != ~0u) {
                        u32 const g = (table_of[k] & 0x0FFFFFFFu) / slot_dwords, chunk = position_of[k] >> 7;
                        u32 const chunks_per_lane = words_per_lane / 4u, part = chunk / chunks_per_lane, within = chunk - part * chunks_per_lane;
                        u32 const entry = (mine[k] * lanes + part) * (queue_pointer_bytes(words_per_lane) / 2u) + within; // 16-bit units
                        u32 *const word = peq + (table_of[k] & 0x0FFFFFFFu) + (entry >> 1);
                        u32 const shift = (entry & 1u) * 16u;
                        if (!((atomicOr(word, 0x8000u << shift) >> shift) & 0x8000u)) atomicOr(word, (atomicAdd(&pool_used[g], 1u) + 1u) << shift);
                    }
                __syncthreads(); // every claimed pointer holds its slot
#pragma unroll
                for (u32 k = 0; k < queue_pattern_reads_k; ++k)
                    if (position_of[k] != ~0u) {
                        u32 const chunk = position_of[k] >> 7;
                        u32 const chunks_per_lane = words_per_lane / 4u, part = chunk / chunks_per_lane, within = chunk - part * chunks_per_lane;
                        u32 const entry = (mine[k] * lanes + part) * (queue_pointer_bytes(words_per_lane) / 2u) + within;
                        u32 const base = table_of[k] & 0x0FFFFFFFu;
                        u32 const slot = (peq[base + (entry >> 1)] >> ((entry & 1u) * 16u)) & 0x7FFFu;
                        atomicOr(&peq[base + pointer_dwords + slot * 4u + ((position_of[k] >> 5) & 3u)], 1u << (position_of[k] & 31u));
                    }
            }
            else {
#pragma unroll
                for (u32 k = 0; k < queue_pattern_reads_k; ++k)
                    if (position_of[k] != ~0u)
                        atomicOr(&peq[(table_of[k] & 0x0FFFFFFFu) + queue_mask_index(table_of[k] >> 28, rows, mine[k], position_of[k] >> 5)],
                                 1u << (position_of[k] & 31u));
            }
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
                queue_table_t const table = {g * slot_dwords, rows, pointer_dwords};
                // ---- a wave block is a chain of dependent steps, one per symbol of its longest candidate, each as many
                //      instructions as the lane holds words: the SIMD's four wavefronts share its issue slots, and a long chain begun
                //      late would hold the launch's end back at a quarter of the speed it could run at.  Longest chain first, on the
                //      SIMD as in the queue: the hardware's own priority, in quarters of the plan's longest chain.
                if (plan.chain_most) {
                    u32 const held = lanes > 1u ? words_per_lane : __builtin_amdgcn_readfirstlane(query.length ? (query.length + 31u) / 32u : 1u);
                    u32 const chain = held * __builtin_amdgcn_readfirstlane(candidates[hi - 1u].length);
                    u32 const quarter = (u32)(((u64)chain * 4u) / plan.chain_most);
                    if (quarter >= 3u) __builtin_amdgcn_s_setprio(3);
                    else if (quarter == 2u) __builtin_amdgcn_s_setprio(2);
                    else if (quarter == 1u) __builtin_amdgcn_s_setprio(1);
                    else __builtin_amdgcn_s_setprio(0);
                }
#define SZS_QUEUE_LANES(W, D) queue_lanes<W, D, runes_>(table, query, candidates, lo, hi, results, results_row_stride, layout)
#define SZS_QUEUE_TEAM(W)                                                                                                          \
    do {                                                                                                                           \
        if (runes_ && sparse) queue_team<W, runes_, runes_>(table, lanes, query, candidates, lo, hi, results, results_row_stride, layout); \
        else queue_team<W, runes_, false>(table, lanes, query, candidates, lo, hi, results, results_row_stride, layout);           \
    } while (0)
                if (lanes > 1u) {
                    switch (words_per_lane) {
                    case 4: SZS_QUEUE_TEAM(4); break;
                    case 8: SZS_QUEUE_TEAM(8); break;
                    case 12: SZS_QUEUE_TEAM(12); break;
                    default: SZS_QUEUE_TEAM(16); break;
                    }
                }
                else {
                    switch (queue_body_words(__builtin_amdgcn_readfirstlane(query.length ? (query.length + 31u) / 32u : 1u), runes_)) {
                    case 1: SZS_QUEUE_LANES(1, 2); break;
                    case 2: SZS_QUEUE_LANES(2, 2); break;
                    case 3: SZS_QUEUE_LANES(3, 2); break;
                    case 4: SZS_QUEUE_LANES(4, 2); break;
                    case 5: SZS_QUEUE_LANES(5, 2); break;
                    case 6: SZS_QUEUE_LANES(6, 2); break;
                    case 7: SZS_QUEUE_LANES(7, 2); break;
                    case 8: SZS_QUEUE_LANES(8, 2); break;
                    case 10:
                        if constexpr (!runes_) SZS_QUEUE_LANES(10, 1);
                        break;
                    case 12: SZS_QUEUE_LANES(12, 1); break;
                    case 16: SZS_QUEUE_LANES(16, 1); break;
                    default:
                        if constexpr (!runes_) SZS_QUEUE_LANES(20, 1);
                        break;
                    }
                }
#undef SZS_QUEUE_LANES
#undef SZS_QUEUE_TEAM
            }
            if (plan.chain_most) __builtin_amdgcn_s_setprio(0);
            } // fit
            done += together;
            if (done >= q_count && tid == 0) next_ticket = ahead - ticket_base;
            __syncthreads(); // C: nobody reads the tables any more; after the last pass the next ticket is visible
        }
    }
    if (trace && tid == 0) {
        u64 *const mine = trace + 7 * (u64)blockIdx.x;
        u64 const ended = wall_clock64();
        if (items_taken && ended - last_began > longest_took) longest_took = ended - last_began, longest_item = last_item;
        mine[0] = began, mine[1] = ended, mine[2] = items_taken, mine[3] = last_item, mine[4] = last_began, mine[5] = longest_item, mine[6] = longest_took;
    }
}

/** Workgroups the device keeps resident (two per CU), per device ordinal and flavour. */
template <bool runes_>
static u32 queue_grid(u64 items) {
    static int resident_of[device_slots_k];
    int *const slot = &resident_of[device_slot()];
    int resident = cached(slot);
    size_t const arena = runes_ ? queue_rune_arena_bytes_k : queue_arena_bytes_k;
    if (!resident) {
        int device = 0, units = 0, per_unit = 0;
        if (hipGetDevice(&device) != hipSuccess ||
            hipDeviceGetAttribute(&units, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<void const *>(levenshtein_myers_queue_kernel<runes_>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)arena) != hipSuccess ||
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_unit, levenshtein_myers_queue_kernel<runes_>, (int)queue_threads_k, arena) != hipSuccess ||
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

extern "C" unsigned szs_hip_levenshtein_myers_queue_grid(uint64_t items, int runes) {
    return runes ? szs_hip::queue_grid<true>(items) : szs_hip::queue_grid<false>(items);
}
extern "C" size_t szs_hip_levenshtein_myers_queue_table_bytes(int runes) {
    return runes ? szs_hip::queue_rune_arena_bytes_k : szs_hip::queue_arena_bytes_k;
}

extern "C" int szs_hip_levenshtein_myers_queue(szs_queue_plan_t const *plan, szs_string_ref_t const *queries,
                                               szs_string_ref_t const *candidates, uint64_t *results, uint64_t results_row_stride,
                                               int layout, uint32_t *tickets, uint32_t ticket_base, uint32_t *tickets_taken,
                                               uint64_t *trace, uint32_t alphabet, uint32_t *unfit_flag, uint32_t unfit_sequence,
                                               void *stream) {
    using namespace szs_hip;
    *tickets_taken = 0;
    if (!plan->items_total) return 0;
    if (plan->tiles_count > SZS_QUEUE_MOST_TILES) return (int)hipErrorInvalidValue;
    hipStream_t const s = static_cast<hipStream_t>(stream);
    u32 grid;
    if (alphabet) {
        grid = queue_grid<true>(plan->items_total);
        hipLaunchKernelGGL(levenshtein_myers_queue_kernel<true>, dim3(grid), dim3(queue_threads_k), queue_rune_arena_bytes_k, s, queries, candidates,
                           results, results_row_stride, layout, tickets, ticket_base, trace, alphabet, unfit_flag, unfit_sequence, *plan);
    }
    else {
        grid = queue_grid<false>(plan->items_total);
        hipLaunchKernelGGL(levenshtein_myers_queue_kernel<false>, dim3(grid), dim3(queue_threads_k), queue_arena_bytes_k, s, queries, candidates,
                           results, results_row_stride, layout, tickets, ticket_base, trace, 0u, unfit_flag, unfit_sequence, *plan);
    }
    hipError_t const error = hipGetLastError();
    if (error == hipSuccess) *tickets_taken = plan->items_total + grid;
    return (int)error;
}
