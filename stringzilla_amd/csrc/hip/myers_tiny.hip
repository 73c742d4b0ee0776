/*
 *  myers_tiny.hip - unit-cost Levenshtein distances over TINY byte strings (words of text: ~6 x 6 cells per pair), straight
 *  from the caller's tapes: no planner, no refs, no sorted sides.
 *
 *  Replaces, for the ROCm build, the reference's tiny-token fast path
 *      unit_myers_singleword_direct_per_cuda_cell_   /root/reference/include/stringzillas/similarities/cuda.cuh:2864
 *      (launch site :4297-4340: one thread per cell, Peq on the fly, straight to the matrix - "removes the host-orchestration
 *      overhead that leaves the GPU >60% idle on tiny-token cross-products")
 *  and returns what levenshtein_distance_myers<char, serial> returns (.../similarities/serial.hpp:2073-2314).
 *
 *  What binds this regime is not the DP (36 cells a pair) but everything around it - and the RESULT MATRIX: 4096 x 4096 words
 *  are 134 MB of 8-byte results against 48 KB of strings, the one place on this path where the HBM roofline is the real one.
 *  The lanes-tier kernels (lev_myers.hip) give every pair a lane and every query a workgroup: ~390 lane-operations per pair
 *  here, and results scattered 8 bytes at a time over the length-sorted columns.  This kernel instead:
 *
 *  - SIXTEEN-BIT bit-vectors, two to a register.  A token of up to 16 bytes is a 16-row pattern; the recurrence runs on both
 *    halves of a 32-bit register at once - the boolean algebra does not care, the one addition is `v_pk_add_u16` (no carry
 *    between the halves) and the two shifts are `v_pk_lshlrev_b16`.  A lane scores its candidate against a GROUP of 2 R queries
 *    at once: R registers of VP, R of VN, R independent dependency chains to interleave (R = 16 or 8: template parameter).
 *  - The masks of a group sit side by side in LDS, `peq[byte][R dwords]`: one row per text byte hands a lane the masks of the
 *    whole group (R / 4 ds_read_b128).  Built with LDS atomics from bytes the threads fetched one group ahead, and un-built (the
 *    same dwords cleared) instead of zeroing the table per group.
 *  - A workgroup owns 256 CONSECUTIVE candidates (a block of result columns) and walks a span of the queries.  Candidates are
 *    only sorted INSIDE the block (a counting sort of 256 lengths in LDS; wavefront w takes the w-th quarter): each wavefront
 *    gets texts of near-equal length and a query's 256 results still form one contiguous 2 KB run of its row.  The candidate's
 *    bytes live in four registers.
 *  - Results go through LDS, a byte each (a distance of two tiny tokens is at most 16): `out[j][candidate of the block]` packs
 *    four queries; then every wavefront writes 512 contiguous bytes per row.
 *
 *  Tokens LONGER than 16 bytes (up to 255: a few per cent of a text's words, a third of its DP cells) are scored by the SAME
 *  workgroups in the SAME launch - see "every token in the one launch" below.  (Scored like the others - a long text in a lane
 *  of the group's columns - one long URL in a wavefront holds its sixty-three neighbours and, through the workgroup's barriers,
 *  the other three wavefronts for ten times their own work: 441 us for 4096 x 4096 words of text.  Round 5's first design
 *  listed them in a pass over the tapes and scored them in a kernel of their own from masks tabled in device memory: four
 *  launches, 100 us; the one launch: 81.)
 */
#include "myers_core.hpp"

namespace szs_hip {

constexpr u32 tiny_block_k = 256;        // candidates per workgroup: one per lane
constexpr u32 tiny_rows_k = 16;          // bytes of a tiny token = rows of its bit-vector
constexpr u32 tiny_most_queries_k = 256; // queries of one workgroup's span (their offsets live in LDS)
// (Mask rows 20 dwords apart instead of 16 - their first banks then spread over sixteen values instead of four - measured the
// same 67.6 us: bank conflicts of the mask reads are not what the launch waits for.)

typedef unsigned short tiny_pk_u16 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ u64 tiny_offset(void const *offsets, u32 wide, u64 index) {
    return wide ? static_cast<u64 const *>(offsets)[index] : (u64) static_cast<u32 const *>(offsets)[index];
}
/**
 *  Where a string lies and how long it is.  A caller's tape says it with offsets i and i + 1; the CODEPOINT twin (round 6) is
 *  launched on the byte strings of rune ids that `utf8_narrow_kernel` (hip/utf8.hip) made of a UTF-8 tape - there entry i is one
 *  word, the length above bit 56 and the place in the narrow buffer below (`szs_tape_t::wide` = 2): strings that do not touch, no
 *  entry past the last string.  `bytes` beyond tiny_longest_k: malformed, not scored.
 */
constexpr u64 tiny_packed_place_k = (1ull << 56) - 1;
struct tiny_extent_t {
    u64 from, bytes;
};
template <bool packed_>
__device__ __forceinline__ tiny_extent_t tiny_extent(void const *offsets, u32 wide, u64 index) {
    if constexpr (packed_) {
        u64 const entry = static_cast<u64 const *>(offsets)[index];
        return {entry & tiny_packed_place_k, entry >> 56};
    }
    else {
        u64 const from = tiny_offset(offsets, wide, index), to = tiny_offset(offsets, wide, index + 1);
        return {from, to >= from ? to - from : ~0ull};
    }
}
/** The same of a query of the span, from the words the workgroup keeps in LDS (offsets i and i + 1, or the packed entry). */
template <bool packed_>
__device__ __forceinline__ u64 tiny_query_from(u64 const *query_offsets, u32 q) {
    return packed_ ? query_offsets[q] & tiny_packed_place_k : query_offsets[q];
}
template <bool packed_>
__device__ __forceinline__ u64 tiny_query_bytes(u64 const *query_offsets, u32 q) {
    if constexpr (packed_) return query_offsets[q] >> 56;
    else return query_offsets[q + 1] >= query_offsets[q] ? query_offsets[q + 1] - query_offsets[q] : ~0ull;
}

__device__ __forceinline__ u32 tiny_pk_add(u32 a, u32 b) { // v_pk_add_u16: the halves do not carry into each other
    tiny_pk_u16 const sum = __builtin_bit_cast(tiny_pk_u16, a) + __builtin_bit_cast(tiny_pk_u16, b);
    return __builtin_bit_cast(u32, sum);
}
__device__ __forceinline__ u32 tiny_pk_shl1(u32 a) { // v_pk_lshlrev_b16
    tiny_pk_u16 const shifted = __builtin_bit_cast(tiny_pk_u16, a) << (tiny_pk_u16)(1);
    return __builtin_bit_cast(u32, shifted);
}

__device__ __forceinline__ void tiny_set_priority(u32 priority) { // (`s_setprio` takes an immediate; `priority` is wave-uniform)
    if (priority == 0) __builtin_amdgcn_s_setprio(0);
    else if (priority == 1) __builtin_amdgcn_s_setprio(1);
    else if (priority == 2) __builtin_amdgcn_s_setprio(2);
    else __builtin_amdgcn_s_setprio(3);
}

/** One DP column of TWO 16-row patterns at once (the column update of myers_core.hpp on packed halves). */
__device__ __forceinline__ void tiny_column(u32 &vp, u32 &vn, u32 eq) {
    u32 const xv = eq | vn;
    u32 const sum = tiny_pk_add(eq & vp, vp);
    u32 const d0 = (sum ^ vp) | eq;
    u32 const hp = vn | ~(d0 | vp);
    u32 const hn = vp & d0;
    u32 const hp_shifted = tiny_pk_shl1(hp) | 0x00010001u; // the constant +1 of DP row zero enters bit 0 of each half
    u32 const hn_shifted = tiny_pk_shl1(hn);
    vp = hn_shifted | ~(xv | hp_shifted);
    vn = hp_shifted & xv;
}

/* ---- EVERY token in the one launch ---------------------------------------------------------------------------------------------
 *
 *  The longer tokens (17 ... 255 bytes) ride along with the workgroups' tiny-token work.  A workgroup meets three kinds of them:
 *
 *    A  a LONG CANDIDATE of its block (sorted behind the tiny ones by the local sort; its lane sits out the group's columns)
 *       against the group's 2 R tiny queries: the text is the same for a CLUSTER of R lanes, lane d of them advancing the two
 *       patterns of dword d of the group's masks - the very LDS rows the group's own columns read - one register each of VP /
 *       VN, a chain of eleven instructions per text byte instead of R registers' worth.  Wavefront w takes the block's long
 *       candidates (64 / R) w ... (then 256 / R further on): nobody waits for a wavefront that happens to hold them all.  The
 *       distances (at most 255: a byte) land in the group's `out` rows, and leave with the tiny ones' in whole runs.
 *    B  a LONG QUERY of its span against the block's tiny candidates: the query becomes an ordinary W-word Myers pattern (W = 1,
 *       2, 4, 8 by the span's longest) whose match masks are tabled in the LDS the groups' masks have left (R / W patterns a
 *       round); every lane runs its own candidate - up to sixteen columns, the bytes still in its registers - over each of
 *       them.  Rows leave through `out` (16 bits a distance) as whole 2 KB runs.
 *    C  long query x long candidate: the clusters of kind A again, lane r of a cluster advancing pattern r of the round.
 *
 *  No list, no table in device memory, nothing to set back.  Malformed offsets, a string beyond 255 bytes, or a block / span of
 *  which more than a quarter is long leave `*unfit = unfit_sequence` (pinned memory) and the host scores the call the ordinary way.
 */
constexpr u32 tiny_longest_k = SZS_TINY_LONGEST; // bytes of the longest string this kernel scores: a distance fits a byte of `out`

/**
 *  A long candidate's bytes, HELD by the R lanes of a cluster: lane e of the cluster keeps the text's dwords e, R + e, 2 R + e ...
 *  (spliced to the text's own alignment; 64 dwords a cluster) - all loads in flight at once, ahead of the columns they feed.  The
 *  walks below then take the next four bytes from the cluster by `ds_bpermute`: no load from memory stands in a chain of
 *  dependent columns.
 */
template <u32 registers_>
struct tiny_held_text_t {
    static constexpr u32 dwords_k = 64u / registers_;
    u32 dwords[dwords_k];
    u32 column, length; // of the block; 0 bytes where the cluster has no text
};
template <u32 registers_>
__device__ __forceinline__ tiny_held_text_t<registers_> tiny_hold_text(szs_tape_t const &candidates, u64 const *froms, u32 const *lengths,
                                                                        u32 const *lane_of_rank, u32 tiny_count, u32 long_count, u32 k) {
    tiny_held_text_t<registers_> held;
    bool const live = k < long_count;
    held.column = live ? lane_of_rank[tiny_count + k] : 0u;
    held.length = live ? lengths[held.column] : 0u;
    text_stream_t const text(candidates.base + froms[held.column], held.length);
    u32 const e = threadIdx.x & (registers_ - 1u);
    constexpr u32 n = tiny_held_text_t<registers_>::dwords_k;
    u32 raw[2 * n];
#pragma unroll
    for (u32 c = 0; c < n; ++c) raw[2 * c] = text.raw(registers_ * c + e), raw[2 * c + 1] = text.raw(registers_ * c + e + 1);
#pragma unroll
    for (u32 c = 0; c < n; ++c) held.dwords[c] = text.splice(raw[2 * c], raw[2 * c + 1]);
    return held;
}
/** Bytes [4 i, 4 i + 4) of the text this lane's cluster holds (`i` the same for the whole wavefront). */
template <u32 registers_>
__device__ __forceinline__ u32 tiny_held_four(tiny_held_text_t<registers_> const &held, u32 i) {
    u32 const chunk = i / registers_;
    u32 mine = held.dwords[0];
#pragma unroll
    for (u32 c = 1; c < tiny_held_text_t<registers_>::dwords_k; ++c) mine = chunk == c ? held.dwords[c] : mine; // (uniform: scalar selects)
    return (u32)__shfl((int)mine, (int)((threadIdx.x & 63u & ~(registers_ - 1u)) + (i & (registers_ - 1u))), 64);
}

/** Kinds B and C: the span's long queries `listed[0 ... listed_count)` as W-word patterns, R / W of them a round. */
template <int words_, u32 registers_, bool packed_>
__device__ __forceinline__ void tiny_long_queries(u32 *peq, u32 *out, u64 const *query_offsets, unsigned short const *listed, u32 listed_count,
                                                  szs_tape_t const &queries, szs_tape_t const &candidates, u32 query_first,
                                                  u64 const *froms, u32 const *lengths, u32 const *lane_of_rank, u32 tiny_count,
                                                  u32 long_count, tiny_held_text_t<registers_> const &held_first, u32 column,
                                                  bool column_is_tiny, u32 text_length, u32 longest_in_wave, u32 const (&symbols)[4],
                                                  bool my_exists, u32 my_candidate, u64 *__restrict__ results, u64 results_row_stride) {
    static_assert(words_ <= (int)registers_, "a pattern's table must fit the LDS of the groups' masks");
    constexpr u32 per_round = registers_ / words_; // tables of 256 rows x W dwords in the R KB of `peq`; rows of 256 x 16 bits in `out`
    constexpr u32 rows_k = 32u * words_;
    constexpr u32 texts_per_wave = 64u / registers_;
    u32 const tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    unsigned short *const out16 = reinterpret_cast<unsigned short *>(out); // [pattern of the round][column of the block]
    auto start = [&](u32 pad, u32 (&vp)[words_], u32 (&vn)[words_]) { // phantom low rows below a pattern shorter than 32 W
#pragma unroll
        for (int w = 0; w < words_; ++w) {
            u32 const first_bit = 32u * w;
            vp[w] = first_bit >= pad ? ~0u : (first_bit + 32u <= pad ? 0u : (~0u << (pad - first_bit)));
            vn[w] = 0;
        }
    };
    auto masks_of = [&](u32 r, u32 byte, u32 (&eq)[words_]) {
#pragma unroll
        for (int w = 0; w < words_; ++w) eq[w] = peq[(r * 256u + byte) * words_ + w];
    };
    auto distance = [&](u32 length, u32 const (&vp)[words_], u32 const (&vn)[words_]) -> u32 {
        u32 sum = length;
#pragma unroll
        for (int w = 0; w < words_; ++w) sum += (u32)__builtin_popcount(vp[w]) - (u32)__builtin_popcount(vn[w]);
        return sum;
    };
#pragma unroll 1
    for (u32 first = 0; first < listed_count; first += per_round) {
        u32 const here = listed_count - first < per_round ? listed_count - first : per_round;
        // ---- the round's tables: thread p ORs bit `pad + p` of pattern r into the row of the pattern's p-th byte (all of the round's
        //      bytes are requested before the first of them is used: one round trip to the L2, not one per pattern)
        u32 bytes_of[per_round];
#pragma unroll
        for (u32 r = 0; r < per_round; ++r) {
            bytes_of[r] = 0x100u;
            if (r < here) {
                u32 const q = listed[first + r];
                u64 const from = tiny_query_from<packed_>(query_offsets, q);
                if (tid < (u32)tiny_query_bytes<packed_>(query_offsets, q)) bytes_of[r] = reinterpret_cast<u8 const *>(queries.base + from)[tid];
            }
        }
#pragma unroll
        for (u32 r = 0; r < per_round; ++r)
            if (bytes_of[r] < 0x100u) {
                u32 const q = listed[first + r];
                u32 const bit = rows_k - (u32)tiny_query_bytes<packed_>(query_offsets, q) + tid;
                atomicOr(&peq[(r * 256u + bytes_of[r]) * words_ + (bit >> 5)], 1u << (bit & 31u));
            }
        __syncthreads();
        // ---- B: this lane's tiny candidate under every pattern of the round
#pragma unroll 1
        for (u32 r = 0; r < here; ++r) {
            u32 const q = listed[first + r];
            u32 const length = (u32)tiny_query_bytes<packed_>(query_offsets, q);
            u32 vp[words_], vn[words_];
            start(rows_k - length, vp, vn);
#pragma unroll
            for (u32 d = 0; d < 4; ++d) {
                if (4 * d >= longest_in_wave) break; // wave-uniform
#pragma unroll
                for (u32 at = 0; at < 4; ++at)
                    if (4 * d + at < text_length) {
                        u32 eq[words_];
                        masks_of(r, (symbols[d] >> (8 * at)) & 0xFFu, eq);
                        myers_column<words_>(vp, vn, eq);
                    }
            }
            if (column_is_tiny) out16[r * 256u + column] = (unsigned short)distance(text_length, vp, vn);
        }
        // ---- C: the block's long candidates, a cluster of R lanes each; lane r of a cluster takes pattern r of the round
        __builtin_amdgcn_s_setprio(3);
#pragma unroll 1
        for (u32 k_first = 0; k_first < long_count; k_first += 4 * texts_per_wave) {
            tiny_held_text_t<registers_> const held =
                k_first ? tiny_hold_text<registers_>(candidates, froms, lengths, lane_of_rank, tiny_count, long_count, k_first + wave * texts_per_wave + lane / registers_)
                        : held_first;
            u32 const longest = (u32)__builtin_amdgcn_readfirstlane(wave_max_u32(held.length));
            u32 const r = lane & (registers_ - 1u);
            bool const live = r < here && held.length;
            u32 const table = r < here ? r : 0u;
            u32 const q = listed[first + table];
            u32 const length = (u32)tiny_query_bytes<packed_>(query_offsets, q);
            u32 vp[words_], vn[words_];
            start(rows_k - length, vp, vn);
            if constexpr (words_ <= 2) { // the next step's masks in flight under this step's columns
                u32 eq_now[4][words_], eq_next[4][words_];
                u32 four_next = tiny_held_four<registers_>(held, 0);
#pragma unroll
                for (u32 i = 0; i < 4; ++i) masks_of(table, (four_next >> (8 * i)) & 0xFFu, eq_now[i]);
                four_next = tiny_held_four<registers_>(held, 1);
#pragma unroll 1
                for (u32 at = 0; at < longest; at += 4) {
#pragma unroll
                    for (u32 i = 0; i < 4; ++i) masks_of(table, (four_next >> (8 * i)) & 0xFFu, eq_next[i]);
                    four_next = tiny_held_four<registers_>(held, at / 4 + 2);
#pragma unroll
                    for (u32 i = 0; i < 4; ++i)
                        if (at + i < held.length) myers_column<words_>(vp, vn, eq_now[i]);
#pragma unroll
                    for (u32 i = 0; i < 4; ++i)
#pragma unroll
                        for (int w = 0; w < words_; ++w) eq_now[i][w] = eq_next[i][w];
                }
            }
            else { // wide patterns (a query of more than 64 bytes in the span): the registers go to VP / VN, only the bytes run ahead
                u32 four_now = tiny_held_four<registers_>(held, 0);
#pragma unroll 1
                for (u32 at = 0; at < longest; at += 4) {
                    u32 const four_next = tiny_held_four<registers_>(held, at / 4 + 1);
#pragma unroll 1
                    for (u32 i = 0; i < 4; ++i)
                        if (at + i < held.length) {
                            u32 eq[words_];
                            masks_of(table, (four_now >> (8 * i)) & 0xFFu, eq);
                            myers_column<words_>(vp, vn, eq);
                        }
                    four_now = four_next;
                }
            }
            if (live) out16[r * 256u + held.column] = (unsigned short)distance(held.length, vp, vn);
        }
        __builtin_amdgcn_s_setprio(1);
        __syncthreads();
        // ---- the rows leave as whole runs (thread t: candidate t of the block); the tables go back to zeros
        if (my_exists) {
#pragma unroll 1
            for (u32 r = 0; r < here; ++r)
                results[(u64)(query_first + listed[first + r]) * results_row_stride + my_candidate] = out16[r * 256u + tid];
        }
        for (u32 i = tid; i < here * 256u * words_; i += 256) peq[i] = 0;
        __syncthreads();
    }
}

template <u32 registers_, bool packed_>
__device__ __forceinline__ void tiny_body(szs_tape_t const &queries, szs_tape_t const &candidates, u32 queries_per_workgroup,
                                                u64 *__restrict__ results, u64 results_row_stride, u32 *unfit, u32 unfit_sequence,
                                                unsigned long long *symbols_out, u64 *rune_totals, u64 *squares_out, u64 *trace, u32 dense) {
    constexpr u32 R = registers_, H = R / 2, group_k = 2 * R; // registers of VP (of VN) a lane, rows of `out`, queries a group
    constexpr u32 texts_per_wave = 64u / R;                   // clusters of R lanes: kinds A and C
    constexpr u32 slots_per_thread = group_k / 16u;           // thread t builds byte t % 16 of slots t / 16 (+ 16)
#define SZS_TINY_STAMP(K) do { if (trace && threadIdx.x == 0) trace[(u64)blockIdx.x * 10 + (K)] = wall_clock64(); } while (0)
    SZS_TINY_STAMP(0);
    __shared__ __attribute__((aligned(16))) u32 peq[256 * R];         // [byte][register]: the group's masks; afterwards the long queries' tables
    // [group parity][j][column of the block], a byte a distance (slots j, j + H, R + j, R + j + H of the group).  TWO of them: a group's
    // rows are written out UNDER THE NEXT GROUP'S COLUMNS, two stores behind every column step - every workgroup of the launch
    // reaches its columns and its stores at about the same time, and with the stores in a phase of their own the device alternated
    // between saturated vector units with idle memory and 67 MB of results draining with idle vector units (tiny tokens alone:
    // 58 us, of which the columns 33 and the stores 7 on top of them; the matrix alone is 23 us of HBM writes)
    __shared__ __attribute__((aligned(16))) u32 out_both[2 * H * tiny_block_k];
    __shared__ u64 query_offsets[tiny_most_queries_k + 1];
    __shared__ u64 froms[tiny_block_k];
    __shared__ u32 lengths[tiny_block_k], bins[32], lane_of_rank[tiny_block_k];
    __shared__ unsigned short listed[tiny_most_queries_k]; // the span's long queries (17 ... 255 bytes), by their place in the span
    __shared__ u32 listed_count, listed_longest, squares_sum;

    u32 const tid = threadIdx.x;
    u32 const blocks = (u32)(((u64)candidates.count + tiny_block_k - 1) / tiny_block_k);
    u32 const block = blockIdx.x % blocks, span = blockIdx.x / blocks;
    u32 const query_first = span * queries_per_workgroup;
    u32 const queries_here = queries.count - query_first < queries_per_workgroup ? queries.count - query_first : queries_per_workgroup;
    if (blockIdx.x == 0 && tid < 2 && symbols_out) { // the call's cell count is the product of these two (the host's profile)
        szs_tape_t const &whole = tid ? candidates : queries;
        if constexpr (packed_) symbols_out[tid] = rune_totals[tid], rune_totals[tid] = 0; // (counted by the pass that made the strings; zero for the next call's)
        else symbols_out[tid] = tiny_offset(whole.offsets, whole.wide, whole.count) - tiny_offset(whole.offsets, whole.wide, 0);
    }
    // ---- the FIRST group's query bytes: requested before anything else, straight from the tape's offsets (two dependent round
    //      trips that would otherwise stand between the local sort and the first masks run beside the candidates' own two)
    u32 const position = tid & 15u;
    u32 ahead[slots_per_thread]; // this thread's byte of its slot(s) of the NEXT group to be built; 0x100: none
#pragma unroll
    for (u32 k = 0; k < slots_per_thread; ++k) {
        u32 const slot = (tid >> 4) + 16 * k;
        ahead[k] = 0x100u;
        if (slot < queries_here) {
            tiny_extent_t const query = tiny_extent<packed_>(queries.offsets, queries.wide, (u64)query_first + slot);
            if (query.bytes <= tiny_rows_k && position < query.bytes) ahead[k] = reinterpret_cast<u8 const *>(queries.base + query.from)[position];
        }
    }
    // ---- once per workgroup: the block's candidates (offsets, local sort by length: tiny ones, then the long ones, then the absent)
    u32 const my_candidate = block * tiny_block_k + tid;
    u64 my_from = 0;
    u32 my_length = 0;
    bool my_exists = false; // this thread's candidate exists and this kernel scores it (up to 255 bytes)
    if (my_candidate < candidates.count) {
        tiny_extent_t const mine = tiny_extent<packed_>(candidates.offsets, candidates.wide, my_candidate);
        my_from = mine.from;
        if (mine.bytes <= tiny_longest_k) my_length = (u32)mine.bytes, my_exists = true;
        else *unfit = unfit_sequence; // malformed, or too long for this kernel: the host scores the call the ordinary way
    }
    for (u32 i = tid; i < queries_here + (packed_ ? 0u : 1u); i += 256)
        query_offsets[i] = packed_ ? static_cast<u64 const *>(queries.offsets)[(u64)query_first + i] : tiny_offset(queries.offsets, queries.wide, (u64)query_first + i);
    for (u32 i = tid; i < 256 * R; i += 256) peq[i] = 0;
    if (tid < 32) bins[tid] = 0;
    if (tid == 0) listed_count = 0, listed_longest = 0, squares_sum = 0;
    froms[tid] = my_from, lengths[tid] = my_exists ? my_length : 0x80000000u;
    __syncthreads();
    u32 const bin = !my_exists ? tiny_rows_k + 2 : my_length <= tiny_rows_k ? my_length : tiny_rows_k + 1;
    u32 const place_in_bin = atomicAdd(&bins[bin], 1u);
    // a SYMMETRIC call (both tapes are the one tape) counts the cells of its lower triangle, ((sum len)^2 + sum len^2) / 2: the blocks of
    // the first span leave the sums of their candidates' squared lengths where the host adds them up (plain stores to pinned memory)
    if (squares_out && span == 0 && my_exists) atomicAdd(&squares_sum, my_length * my_length);
    for (u32 i = tid; i < queries_here; i += 256) { // the span's long queries, in whatever order the atomics hand out
        u64 const bytes = tiny_query_bytes<packed_>(query_offsets, i);
        if (bytes > tiny_longest_k) *unfit = unfit_sequence;
        else if (bytes > tiny_rows_k) listed[atomicAdd(&listed_count, 1u)] = (unsigned short)i, atomicMax(&listed_longest, (u32)bytes);
    }
    __syncthreads();
    if (tid < 32) { // exclusive scan of the bins by half a wavefront
        u32 const mine = bins[tid];
        u32 inclusive = mine;
#pragma unroll
        for (int offset = 1; offset < 32; offset <<= 1) {
            u32 const other = (u32)__shfl_up((int)inclusive, offset, 64);
            if (tid >= (u32)offset) inclusive += other;
        }
        bins[tid] = inclusive - mine;
    }
    __syncthreads();
    lane_of_rank[bins[bin] + place_in_bin] = tid;
    if (squares_out && span == 0 && tid == 0) squares_out[block] = squares_sum;
    __syncthreads();
    SZS_TINY_STAMP(1);
    u32 const tiny_count = bins[tiny_rows_k + 1], long_count = bins[tiny_rows_k + 2] - tiny_count; // of the block's candidates
    // Long strings ride along as long as they are FEW: a block or a span where more than a quarter is long is not a batch of tiny
    // tokens (sentences, say, under the counts of the previous call's words) - every long string costs its workgroup a chain of
    // dependent columns, and such a call would be scored correctly but many times slower than by the ordinary kernels.  Refused:
    // the whole workgroup leaves, the host scores the call the ordinary way.
    if (!dense && (long_count > tiny_block_k / 4 || listed_count > (queries_here / 4 > 4 ? queries_here / 4 : 4))) {
        if (tid == 0) *unfit = unfit_sequence;
        return;
    }
    // this lane SCORES the candidate of rank `tid` - column `column` of the block: wavefront w the w-th quarter by length.  (The
    // hardware starts the wavefronts of successive workgroups on successive SIMDs, so the four resident workgroups' longest
    // quarters already sit on four different SIMDs; a rotation by the workgroup's place on its CU on top of that undoes it -
    // measured, tiny tokens alone: 56 us as it is, 76 with every workgroup's longest quarter on the same SIMD.)
    u32 const column = lane_of_rank[tid];
    bool const column_is_tiny = lengths[column] <= tiny_rows_k;
    u32 const text_length = column_is_tiny ? lengths[column] : 0u; // a long or absent column consumes nothing in the groups' columns

    text_stream_t const text(candidates.base + froms[column], text_length);
    u32 symbols[4]; // the text's (up to) 16 bytes
    {
        u32 raw[5];
#pragma unroll
        for (u32 d = 0; d < 5; ++d) raw[d] = text.raw(d);
#pragma unroll
        for (u32 d = 0; d < 4; ++d) symbols[d] = text.splice(raw[d], raw[d + 1]);
    }
    u32 const longest_in_wave = (u32)__builtin_amdgcn_readfirstlane(wave_max_u32(text_length));
    u32 const lane = tid & 63u, wave = tid >> 6;
    // the block's first long candidates, 64 / R to a wavefront, held by clusters of R lanes for kinds A and C
    tiny_held_text_t<R> const held_first = tiny_hold_text<R>(candidates, froms, lengths, lane_of_rank, tiny_count, long_count, wave * texts_per_wave + lane / R);
    SZS_TINY_STAMP(2);

    auto length_of = [&](u32 query) -> u32 { // of a query of the span; 0 past the span's end, ~0 for one the groups skip
        if (query >= queries_here) return 0;
        u64 const bytes = tiny_query_bytes<packed_>(query_offsets, query);
        return bytes <= tiny_rows_k ? (u32)bytes : ~0u;
    };
    auto fetch = [&](u32 query) -> u32 { // this thread's byte of that query, 0x100 where it has none
        u32 const length = length_of(query);
        if (length == ~0u || position >= length) return 0x100u;
        return reinterpret_cast<u8 const *>(queries.base + tiny_query_from<packed_>(query_offsets, query))[position];
    };
    auto start_of = [&](u32 low, u32 high) -> u32 { // VP of a register whose halves hold patterns of `low` and `high` rows (~0: none)
        u32 const low_rows = low == ~0u ? 0u : low, high_rows = high == ~0u ? 0u : high;
        return ((0xFFFFu << (tiny_rows_k - low_rows)) & 0xFFFFu) | ((0xFFFF0000u << (tiny_rows_k - high_rows)) & 0xFFFF0000u);
    };
    u32 parity = 0, pending = 0, pending_skipped = 0, pending_first = 0; // the group whose rows still wait in out_both[parity ^ 1]
    u64 *const my_results = results + my_candidate;
    auto flush = [&](u32 step) { // two of the waiting group's rows for this thread's candidate: bytes 2 (step & 1), + 1 of row step / 2 of its `out`
        if (!pending || !my_exists) return;
        u32 const j = step >> 1;
        u32 const packed = out_both[(parity ^ 1u) * H * tiny_block_k + j * tiny_block_k + tid];
#pragma unroll
        for (u32 b = 2 * (step & 1u); b < 2 * (step & 1u) + 2; ++b) {
            u32 const s = j + H * b;
            if (!((pending_skipped >> s) & 1u)) my_results[(u64)(query_first + pending_first + s) * results_row_stride] = (packed >> (8 * b)) & 0xFFu;
        }
    };
#pragma unroll 1
    for (u32 group_first = 0; group_first < queries_here; group_first += group_k) {
        u32 *const out = out_both + parity * H * tiny_block_k;
        // (the text's bytes are the same for every group; left alone, the compiler keeps the sixteen LDS row addresses they select in
        // sixteen registers across the whole loop - two instructions a column step recompute one)
        asm volatile("" : "+v"(symbols[0]), "+v"(symbols[1]), "+v"(symbols[2]), "+v"(symbols[3]));
        // Wave priority by PROGRESS: a workgroup on its first group runs above one on its second, and so on.  The four workgroups
        // of a CU start together, and the hardware serves the oldest wavefront first: left alone, the first of them finished in 27 us,
        // the fourth in 50 (tiny tokens alone), and the CU ran on three, two, one workgroup for the second half of the launch -
        // bound by latency, with less and less to hide it behind.  Levelled by progress they end together: 67 -> 63 us.
        u32 const priority = group_first / group_k < 3u ? 3u - group_first / group_k : 0u;
        tiny_set_priority(priority);
        // masks: slot s lives in half s / R of dword s % R of every row; its query is right-aligned in the half's sixteen bits
        u32 built[slots_per_thread];
#pragma unroll
        for (u32 k = 0; k < slots_per_thread; ++k) {
            u32 const slot = (tid >> 4) + 16 * k;
            u32 const length = length_of(group_first + slot);
            if (ahead[k] < 0x100u) atomicOr(&peq[ahead[k] * R + (slot % R)], ((slot / R) ? 0x10000u : 1u) << (tiny_rows_k - length + position));
            built[k] = ahead[k];
            ahead[k] = fetch(group_first + group_k + slot); // the next group's bytes go out now and come back under the scoring below
        }
        // the group's 2 R lengths: lane l of every wavefront works out slot l's, `readlane` hands them round as scalars
        u32 const length_of_my_slot = length_of(group_first + (lane & (group_k - 1u)));
        __syncthreads();
        if (group_first == 0) SZS_TINY_STAMP(3);
        u32 skipped = 0; // bit s: slot s is a long query (kind B's) or lies past the span's end
        {
            u32 vp[R], vn[R];
#pragma unroll
            for (u32 d = 0; d < R; ++d) {
                u32 const low = __builtin_amdgcn_readlane(length_of_my_slot, d), high = __builtin_amdgcn_readlane(length_of_my_slot, d + R);
                skipped |= (low == ~0u || group_first + d >= queries_here ? 1u : 0u) << d;
                skipped |= (high == ~0u || group_first + d + R >= queries_here ? 1u : 0u) << (d + R);
                vp[d] = start_of(low, high), vn[d] = 0;
            }
            uint4 const *const rows = reinterpret_cast<uint4 const *>(peq);
            auto take = [&](u32 symbol) {
                uint4 const *const row = rows + symbol * (R / 4);
                u32 masks[R];
#pragma unroll
                for (u32 c = 0; c < R / 4; ++c) {
                    uint4 const four = row[c];
                    masks[4 * c] = four.x, masks[4 * c + 1] = four.y, masks[4 * c + 2] = four.z, masks[4 * c + 3] = four.w;
                }
#pragma unroll
                for (u32 d = 0; d < R; ++d) tiny_column(vp[d], vn[d], masks[d]);
            };
            u32 const steps_here = (longest_in_wave + 3u) & ~3u; // wave-uniform: the column steps this wavefront walks
#pragma unroll
            for (u32 d = 0; d < 4; ++d) {
                if (4 * d >= steps_here) break;
#pragma unroll
                for (u32 at = 0; at < 4; ++at) {
                    if (4 * d + at < text_length) take((symbols[d] >> (8 * at)) & 0xFFu);
                    if (4 * d + at < R) flush(4 * d + at); // the previous group's rows leave under these columns
                }
            }
#pragma unroll
            for (u32 step = 0; step < R; ++step)
                if (step >= steps_here) flush(step); // (a wavefront of short texts: the rest of them now)
            // ---- results: distance = text length + popcount(VP) - popcount(VN) per half (phantom rows hold zeros; at most 16, a byte
            //      each), through LDS so that every wavefront writes 512 contiguous bytes of a row.  (Written straight from the
            //      registers - a lane its column, row by row - the launch took 70 us instead of 67: partial lines.)
            if (column_is_tiny) { // (a long column's bytes of `out` are kind A's to write)
#pragma unroll
                for (u32 j = 0; j < H; ++j) {
                    u32 packed = 0;
#pragma unroll
                    for (u32 k = 0; k < 2; ++k) {
                        u32 const d = j + H * k;
                        u32 const low = text_length + (u32)__builtin_popcount(vp[d] & 0xFFFFu) - (u32)__builtin_popcount(vn[d] & 0xFFFFu);
                        u32 const high = text_length + (u32)__builtin_popcount(vp[d] >> 16) - (u32)__builtin_popcount(vn[d] >> 16);
                        packed |= (low << (8 * k)) | (high << (16 + 8 * k));
                    }
                    out[j * tiny_block_k + column] = packed;
                }
            }
        }
        if (group_first == 0) SZS_TINY_STAMP(4);
        // ---- A: the block's long candidates under the group's masks - R lanes a text, lane d of them the patterns of dword d.
        //      A chain of dependent columns as long as the text: the wavefront runs it at a raised priority (`s_setprio`) - few
        //      instructions, but the launch ends when the workgroup with the longest chains does.
        if (long_count) __builtin_amdgcn_s_setprio(3);
#pragma unroll 1
        for (u32 k_first = 0; k_first < long_count; k_first += 4 * texts_per_wave) {
            tiny_held_text_t<R> const held =
                k_first ? tiny_hold_text<R>(candidates, froms, lengths, lane_of_rank, tiny_count, long_count, k_first + wave * texts_per_wave + lane / R) : held_first;
            u32 const d = lane & (R - 1u);
            u32 const longest = (u32)__builtin_amdgcn_readfirstlane(wave_max_u32(held.length));
            u32 vp = start_of(length_of(group_first + d), length_of(group_first + d + R)), vn = 0;
            u32 masks_now[4], masks_next[4];
            u32 four_next = tiny_held_four<R>(held, 0);
#pragma unroll
            for (u32 i = 0; i < 4; ++i) masks_now[i] = peq[((four_next >> (8 * i)) & 0xFFu) * R + d];
            four_next = tiny_held_four<R>(held, 1);
#pragma unroll 1
            for (u32 at = 0; at < longest; at += 4) {
#pragma unroll
                for (u32 i = 0; i < 4; ++i) masks_next[i] = peq[((four_next >> (8 * i)) & 0xFFu) * R + d]; // the next step's masks ...
                four_next = tiny_held_four<R>(held, at / 4 + 2);                                            // ... and the bytes of the one after
#pragma unroll
                for (u32 i = 0; i < 4; ++i)
                    if (at + i < held.length) tiny_column(vp, vn, masks_now[i]);
#pragma unroll
                for (u32 i = 0; i < 4; ++i) masks_now[i] = masks_next[i];
            }
            if (held.length) {
                u8 *const bytes = reinterpret_cast<u8 *>(out) + ((u64)((d % H) * tiny_block_k + held.column)) * 4u + (d / H);
                bytes[0] = (u8)(held.length + (u32)__builtin_popcount(vp & 0xFFFFu) - (u32)__builtin_popcount(vn & 0xFFFFu));
                bytes[2] = (u8)(held.length + (u32)__builtin_popcount(vp >> 16) - (u32)__builtin_popcount(vn >> 16));
            }
        }
        tiny_set_priority(priority);
        __syncthreads();
        if (group_first == 0) SZS_TINY_STAMP(5);
        // un-build the masks (the same dwords back to zero: cheaper than clearing the table) ...
#pragma unroll
        for (u32 k = 0; k < slots_per_thread; ++k)
            if (built[k] < 0x100u) peq[built[k] * R + (((tid >> 4) + 16 * k) % R)] = 0;
        // ... and the rows wait for the next group's columns (the rows of long queries are kind B's)
        pending = 1, pending_skipped = skipped, pending_first = group_first, parity ^= 1u;
        if (group_first == 0) SZS_TINY_STAMP(6);
        __syncthreads(); // the next group's atomics must not meet the un-building stores, nor its distances these reads
    }
#pragma unroll
    for (u32 step = 0; step < R; ++step) flush(step); // the last group's rows
    __builtin_amdgcn_s_setprio(1); // (what is left - the span's long queries - yields to workgroups still on their groups)
    SZS_TINY_STAMP(7);
    // ---- B and C: the span's long queries, as W-word patterns by the longest of them
    u32 *const out = out_both;
    u32 const long_queries = listed_count;
    if (long_queries) __syncthreads(); // (their rows are staged where the last rows may just have been read)
    if (long_queries) {
        u32 const longest_query = listed_longest;
#define SZS_TINY_LONG(W)                                                                                                                   \
    tiny_long_queries<W, R, packed_>(peq, out, query_offsets, listed, long_queries, queries, candidates, query_first, froms, lengths, lane_of_rank, \
                            tiny_count, long_count, held_first, column, column_is_tiny, text_length, longest_in_wave, symbols, my_exists,  \
                            my_candidate, results, results_row_stride)
        if (longest_query <= 32) SZS_TINY_LONG(1);
        else if (longest_query <= 64) SZS_TINY_LONG(2);
        else if (longest_query <= 128) SZS_TINY_LONG(4);
        else SZS_TINY_LONG(8);
#undef SZS_TINY_LONG
    }
    SZS_TINY_STAMP(8);
#undef SZS_TINY_STAMP
}

/*  Measured and not kept, round 6 (profiles/r06/tiny_split_long_queries.txt): the long queries of every (block, span) in a workgroup
 *  of their own - the grid doubled, even workgroups the groups, odd ones kinds B and C (they share nothing but their set-up, and the
 *  long queries would no longer trail a workgroup's last group).  Words of text 96 -> 124 us, tokens of exactly 16 bytes - no long query
 *  anywhere, the odd workgroups leave after their set-up - 67 -> 124 us: the launch lasts about as long as ONE workgroup lives (57 us on
 *  average) because all 1024 are resident at once; 2048 workgroups are two rounds of residents, and a workgroup does not live shorter
 *  for having fewer neighbours.  What binds this launch is the latency of a workgroup's own chain of phases, not the device's throughput.
 *
 *  Measured and not kept (4096 x 4096 words of text, 81 us as it stands):
 *  - the groups DRAWN from a ticket counter per block of candidates (a workgroup its own place first, then whatever comes; the next
 *    group's bytes and lengths requested straight from the tape one group ahead, the ticket two ahead): 82 us.  A workgroup lives
 *    54 us on average and the launch ends at 76 - but what the last workgroups are busy with is not groups somebody else could take:
 *    it is the long queries of their own groups (kinds B and C, 9 us on average and up to 28 behind the last group) and the long
 *    candidates of their own block.  (With all blocks' counters in one 128-byte line the thousand first draws queued up at one
 *    place in the L2: 12.8 us to the first barrier instead of 3.5.)
 *  - kinds B and C at the top wave priority instead of below the groups: the same.
 *  - half of the workgroups starting 4 ... 12 us late, so that some compute while others drain: the late half runs 6 us faster, and
 *    the launch ends 4 ... 5 us later.
 */
/** R = 16: thirty-two queries a group, four wavefronts a SIMD (128 registers), 39 KB of LDS - four workgroups a CU.  (R = 8 - groups
 *  of sixteen, 64 or 80 registers for eight or six wavefronts a SIMD - measured 66 ... 75 us on tiny tokens alone where this takes
 *  56: twice the mask builds, barriers and staging per pair, and what the launch waits for is not hidden by more wavefronts.) */
template <bool packed_>
__global__ __launch_bounds__(256, 4) void levenshtein_tiny_kernel(szs_tape_t queries, szs_tape_t candidates, u32 queries_per_workgroup,
                                                                       u64 *__restrict__ results, u64 results_row_stride, u32 *unfit,
                                                                       u32 unfit_sequence, unsigned long long *symbols_out, u64 *rune_totals,
                                                                       u64 *squares_out, u64 *trace, u32 dense) {
    tiny_body<16, packed_>(queries, candidates, queries_per_workgroup, results, results_row_stride, unfit, unfit_sequence, symbols_out, rune_totals,
                           squares_out, trace, dense);
}

} // namespace szs_hip

extern "C" int szs_hip_levenshtein_tiny(szs_tape_t const *queries_tape, szs_tape_t const *candidates_tape, uint64_t *results,
                                              uint64_t results_row_stride, uint32_t *unfit, uint32_t unfit_sequence,
                                              unsigned long long *symbols_out, uint64_t *rune_totals, uint64_t *squares_out,
                                              uint64_t *trace, uint64_t trace_workgroups, int dense, void *stream) {
    using namespace szs_hip;
    szs_tape_t const queries = *queries_tape, candidates = *candidates_tape;
    u32 const queries_count = queries.count, candidates_count = candidates.count;
    if (!queries_count || !candidates_count) return 0;
    u32 const group = 32u;
    u64 const blocks = ((u64)candidates_count + tiny_block_k - 1) / tiny_block_k;
    // Spans of the queries: enough workgroups to fill the device a few times over (a workgroup's set-up - offsets, the local sort -
    // is paid once per span), whole groups, at most tiny_most_queries_k queries each.
    // (measured on 4096 x 4096 words: 2048 workgroups of one group each 55.6 us, 1024 of two 50.5, 688 of three 63.0 - a workgroup's
    // set-up is 7 of its 20 us, but fewer workgroups than 4 per CU leave nobody to run while the others wait)
    u64 const wanted_workgroups = 1024;
    u64 spans = (wanted_workgroups + blocks - 1) / blocks;
    u64 per_span = ((u64)queries_count + spans - 1) / spans;
    per_span = (per_span + group - 1) / group * group;
    if (per_span > tiny_most_queries_k) per_span = tiny_most_queries_k;
    spans = ((u64)queries_count + per_span - 1) / per_span;
    if (blocks * spans > 0x7FFFFFFFull) return (int)hipErrorInvalidValue;
    if (blocks * spans > trace_workgroups) trace = nullptr; // the stamps are indexed by workgroup: a grid beyond the buffer is not traced
    bool const packed = queries.wide == 2; // the codepoint twin: both sides are strings of rune ids made by utf8_narrow_kernel
    if (packed != (candidates.wide == 2) || (packed && !rune_totals)) return (int)hipErrorInvalidValue;
    if (packed)
        hipLaunchKernelGGL(levenshtein_tiny_kernel<true>, dim3((u32)(blocks * spans)), dim3(256), 0, static_cast<hipStream_t>(stream), queries, candidates,
                           (u32)per_span, results, results_row_stride, unfit, unfit_sequence, symbols_out, rune_totals, squares_out, trace, dense ? 1u : 0u);
    else
        hipLaunchKernelGGL(levenshtein_tiny_kernel<false>, dim3((u32)(blocks * spans)), dim3(256), 0, static_cast<hipStream_t>(stream), queries, candidates,
                           (u32)per_span, results, results_row_stride, unfit, unfit_sequence, symbols_out, rune_totals, squares_out, trace, dense ? 1u : 0u);
    return (int)hipGetLastError();
}
