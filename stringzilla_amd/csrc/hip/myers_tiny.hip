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
 *    between the halves) and the two shifts are `v_pk_lshlrev_b16`.  A lane scores its candidate against THIRTY-TWO queries at
 *    once: sixteen registers of VP, sixteen of VN, sixteen independent dependency chains to interleave.
 *  - The masks of a group of 32 queries sit side by side in LDS, `peq[byte][16 dwords]`: one 64-byte row per text byte hands a
 *    lane the masks of all thirty-two (four ds_read_b128).  Built with LDS atomics from bytes the threads fetched one group
 *    ahead, and un-built (the same dwords cleared) instead of zeroing 16 KB per group.
 *  - A workgroup owns 256 CONSECUTIVE candidates (a block of result columns) and walks a span of the queries.  Candidates are
 *    only sorted INSIDE the block (a counting sort of 256 lengths in LDS; wavefront w of workgroup b takes the (w + b) % 4-th
 *    quarter, so that the longest quarter does not always land on the same SIMD): each wavefront gets texts of near-equal length
 *    and a query's 256 results still form one contiguous 2 KB run of its row.  The candidate's bytes live in four registers.
 *  - Results go through LDS, a byte each (a distance of two tiny tokens is at most 16): `out[j][candidate of the block]` packs
 *    four queries; then every wavefront writes 512 contiguous bytes per row.
 *
 *  Tokens LONGER than 16 bytes are not this kernel's business: it skips them (their rows / columns are left untouched).
 *  `levenshtein_tiny_prepare_kernel` lists them up front - refs in device memory, at most SZS_TINY_MOST_OUTLIERS per side - for
 *  `levenshtein_outliers_kernel` (lev_myers.hip), which runs BESIDE this one on a second stream.  One long URL in a wavefront
 *  would otherwise hold its sixty-three neighbours - and, through the workgroup's barriers, the other three wavefronts - for ten
 *  times their own work (measured: 441 us for 4096 x 4096 words of text with every token scored here, whatever its length).
 *  More outliers than the list holds, a listed string beyond 256 bytes, malformed offsets: `*unfit = unfit_sequence` (pinned
 *  memory) and the host scores the call the ordinary way.
 */
#include "myers_core.hpp"

namespace szs_hip {

constexpr u32 tiny_group_k = 32;         // queries scored side by side by one lane: two per register
constexpr u32 tiny_block_k = 256;        // candidates per workgroup: one per lane
constexpr u32 tiny_rows_k = 16;          // bytes of a tiny token = rows of its bit-vector
constexpr u32 tiny_most_queries_k = 256; // queries of one workgroup's span (their offsets live in LDS)
// (Rows 20 dwords apart - their first banks then spread over sixteen values instead of four - measured the same 67.6 us: bank
// conflicts of the mask reads are not what the launch waits for.)
constexpr u32 tiny_row_dwords_k = 16;

typedef unsigned short tiny_pk_u16 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ u64 tiny_offset(void const *offsets, u32 wide, u64 index) {
    return wide ? static_cast<u64 const *>(offsets)[index] : (u64) static_cast<u32 const *>(offsets)[index];
}
__device__ __forceinline__ u32 tiny_pk_add(u32 a, u32 b) { // v_pk_add_u16: the halves do not carry into each other
    tiny_pk_u16 const sum = __builtin_bit_cast(tiny_pk_u16, a) + __builtin_bit_cast(tiny_pk_u16, b);
    return __builtin_bit_cast(u32, sum);
}
__device__ __forceinline__ u32 tiny_pk_shl1(u32 a) { // v_pk_lshlrev_b16
    tiny_pk_u16 const shifted = __builtin_bit_cast(tiny_pk_u16, a) << (tiny_pk_u16)(1);
    return __builtin_bit_cast(u32, shifted);
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

/**
 *  Ahead of the two scoring kernels: one thread per string of either tape.  Strings of more than 16 bytes are listed for the
 *  outliers' kernel; malformed offsets, a listed string beyond 256 bytes or more of them than the list holds raise `*unfit`.
 *  (The tiny kernel's own workgroups listed them at first; but then the outliers' kernel could only start when the tiny kernel
 *  had ended, and the two - both bound by latency, not by work - took 55 + 49 us one after the other.  Listed up front, they run
 *  side by side on two streams.)
 */
__global__ __launch_bounds__(256) void levenshtein_tiny_prepare_kernel(szs_tape_t queries, szs_tape_t candidates, u32 query_workgroups,
                                                                      u32 *unfit, u32 unfit_sequence, szs_tiny_outliers_t *outliers,
                                                                      u32 *query_masks, u32 *candidate_masks,
                                                                      unsigned long long *symbols_out, int unbuild) {
    int const side = blockIdx.x < query_workgroups ? 0 : 1;
    szs_tape_t const &tape = side ? candidates : queries;
    u32 const index = (side ? blockIdx.x - query_workgroups : blockIdx.x) * 256 + threadIdx.x;
    if (blockIdx.x == 0 && threadIdx.x < 2 && symbols_out && !unbuild) { // the call's cell count is the product of these two (the host's profile)
        szs_tape_t const &whole = threadIdx.x ? candidates : queries;
        symbols_out[threadIdx.x] = tiny_offset(whole.offsets, whole.wide, whole.count) - tiny_offset(whole.offsets, whole.wide, 0);
    }
    if (index >= tape.count) return;
    u64 const from = tiny_offset(tape.offsets, tape.wide, index), to = tiny_offset(tape.offsets, tape.wide, (u64)index + 1);
    if (to < from) { // malformed offsets: the host's planner reports them
        if (!unbuild) *unfit = unfit_sequence;
        return;
    }
    if (to - from <= tiny_rows_k) { // a tiny string: its match masks, for the outliers' kernel (kernels.h: SZS_TINY_TABLE_BYTES)
        u32 *const masks = side ? candidate_masks : query_masks;
        u32 const rows = (u32)(to - from), row_dwords = ((tape.count + 127u) / 128u) * 64u;
        u32 const dword = (index / 128u) * 64u + (index % 64u), half = (index % 128u) / 64u;
        text_stream_t const text(tape.base + from, rows);
        u32 raw[5], symbols[4]; // all of the string in one round trip; then the atomics
#pragma unroll
        for (u32 d = 0; d < 5; ++d) raw[d] = text.raw(d);
#pragma unroll
        for (u32 d = 0; d < 4; ++d) symbols[d] = text.splice(raw[d], raw[d + 1]);
#pragma unroll
        for (u32 at = 0; at < tiny_rows_k; ++at) {
            if (at >= rows) break;
            u32 *const where = &masks[(u64)((symbols[at / 4] >> (8 * (at % 4))) & 0xFFu) * row_dwords + dword];
            // `unbuild`: the same dwords back to zero when the call's kernels have read them - the tables are all zeros between calls,
            // which spares every call a fill of 512 bytes per string (4 MB and 7.5 us for 4096 x 4096 words)
            if (unbuild) *where = 0;
            else atomicOr(where, (half ? 0x10000u : 1u) << (tiny_rows_k - rows + at));
        }
        return;
    }
    if (unbuild) return;
    u32 const place = atomicAdd(&outliers->counts[side], 1u);
    if (place >= SZS_TINY_MOST_OUTLIERS) { // too many: the host scores the call the ordinary way (the other kernel reads no further)
        *unfit = unfit_sequence;
        return;
    }
    if (to - from > SZS_TINY_LONG_OUTLIER) atomicAdd(&outliers->long_counts[side], 1u);
    szs_string_ref_t ref;
    ref.address = tape.base + from, ref.length = (u32)(to - from), ref.index = index;
    if (to - from > 32u * SZS_MYERS_SHORT_WORDS) // too long for the outliers' kernel's bodies: listed as an EMPTY string (every slot that
        ref.length = 0, *unfit = unfit_sequence; // kernel reads holds a ref of this call), and the call is scored again
    outliers->refs[side][place] = ref;
}

__global__ __launch_bounds__(256) void levenshtein_tiny_kernel(szs_tape_t queries, szs_tape_t candidates, u32 queries_per_workgroup,
                                                              u64 *__restrict__ results, u64 results_row_stride, u64 *trace) {
#define SZS_TINY_STAMP(K) do { if (trace && threadIdx.x == 0) trace[(u64)blockIdx.x * 8 + (K)] = wall_clock64(); } while (0)
    SZS_TINY_STAMP(0);
    __shared__ __attribute__((aligned(16))) u32 peq[256 * tiny_row_dwords_k]; // [byte][dword d: slots d (low half) and d + 16 (high)]: 20 KB
    __shared__ u32 out[8 * tiny_block_k]; // [j][column of the block]: the distances of slots j, j + 8, j + 16, j + 24, a byte each: 8 KB
    __shared__ u64 query_offsets[tiny_most_queries_k + 1];
    __shared__ u64 froms[tiny_block_k];
    __shared__ u32 lengths[tiny_block_k], bins[32], lane_of_rank[tiny_block_k];

    u32 const tid = threadIdx.x;
    u32 const blocks = (candidates.count + tiny_block_k - 1) / tiny_block_k;
    u32 const block = blockIdx.x % blocks, span = blockIdx.x / blocks;
    u32 const query_first = span * queries_per_workgroup;
    u32 const queries_here = queries.count - query_first < queries_per_workgroup ? queries.count - query_first : queries_per_workgroup;
    // ---- the FIRST group's query bytes: requested before anything else, straight from the tape's offsets (not through the LDS copy
    //      of them, a barrier away): two dependent round trips that used to stand between the local sort and the first masks
    //      (4.6 of a workgroup's 20 us) now run beside the candidates' own two
    u32 first_low = 0x100u, first_high = 0x100u;
    {
        u32 const at = tid & 15u;
        u32 const slots[2] = {tid >> 4, (tid >> 4) + 16};
        u64 from[2] = {0, 0}, to[2] = {0, 0};
#pragma unroll
        for (u32 k = 0; k < 2; ++k)
            if (slots[k] < queries_here)
                from[k] = tiny_offset(queries.offsets, queries.wide, (u64)query_first + slots[k]), to[k] = tiny_offset(queries.offsets, queries.wide, (u64)query_first + slots[k] + 1);
        if (to[0] >= from[0] && to[0] - from[0] <= tiny_rows_k && at < to[0] - from[0]) first_low = reinterpret_cast<u8 const *>(queries.base + from[0])[at];
        if (to[1] >= from[1] && to[1] - from[1] <= tiny_rows_k && at < to[1] - from[1]) first_high = reinterpret_cast<u8 const *>(queries.base + from[1])[at];
    }
    // ---- once per workgroup: the block's candidates (offsets, local sort by length), the span's query offsets, clean masks
    u32 const my_candidate = block * tiny_block_k + tid;
    u64 my_from = 0;
    u32 my_length = 0;
    bool my_tiny = false; // this thread's candidate exists and is scored here (up to 16 bytes)
    if (my_candidate < candidates.count) {
        my_from = tiny_offset(candidates.offsets, candidates.wide, my_candidate);
        u64 const to = tiny_offset(candidates.offsets, candidates.wide, (u64)my_candidate + 1);
        if (to >= my_from && to - my_from <= tiny_rows_k) my_length = (u32)(to - my_from), my_tiny = true; // (longer: the outliers' kernel's)
    }
    for (u32 i = tid; i <= queries_here; i += 256) query_offsets[i] = tiny_offset(queries.offsets, queries.wide, (u64)query_first + i);
    for (u32 i = tid; i < 256 * tiny_row_dwords_k; i += 256) peq[i] = 0;
    if (tid < 32) bins[tid] = 0;
    froms[tid] = my_from, lengths[tid] = my_tiny ? my_length : 0x80000000u; // (a skipped column sorts last and scores as an empty text)
    __syncthreads();
    // counting sort of the block's 256 lengths (0 ... 16, skipped columns last): rank -> the thread that holds that candidate
    u32 const bin = my_tiny ? my_length : tiny_rows_k + 1;
    u32 const place_in_bin = atomicAdd(&bins[bin], 1u);
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
    __syncthreads();
    SZS_TINY_STAMP(1);
    // this lane SCORES the candidate of rank ((wave + block) % 4) x 64 + lane - column `column` of the block
    u32 const column = lane_of_rank[(((tid >> 6) + blockIdx.x) & 3u) * 64u + (tid & 63u)];
    u32 const text_length = lengths[column] & 0x7FFFFFFFu; // 0 for a skipped column: nothing to consume, nothing written

    text_stream_t const text(candidates.base + froms[column], text_length);
    u32 symbols[4]; // the text's (up to) 16 bytes
    {
        u32 raw[5];
#pragma unroll
        for (u32 d = 0; d < 5; ++d) raw[d] = text.raw(d);
#pragma unroll
        for (u32 d = 0; d < 4; ++d) symbols[d] = text.splice(raw[d], raw[d + 1]);
    }
    u32 const longest_in_wave = (u32)__builtin_amdgcn_readfirstlane(wave_max_u32(text_length)); // (the compiler must know it is uniform)
    SZS_TINY_STAMP(2);

    // ---- the span's queries, thirty-two at a time.  Thread t holds byte (t % 16) of slot t / 16 and of slot 16 + t / 16 - both
    //      live in dword t / 16 of a row: the bytes of the NEXT group are in flight while this one is scored.
    u32 const dword_of_mine = tid >> 4, position = tid & 15u, lane = tid & 63u;
    auto length_of = [&](u32 query) -> u32 { // of a query of the span; 0 past the span's end, ~0 for one this kernel skips
        if (query >= queries_here) return 0;
        u64 const from = query_offsets[query], to = query_offsets[query + 1];
        return to >= from && to - from <= tiny_rows_k ? (u32)(to - from) : ~0u;
    };
    auto fetch = [&](u32 query) -> u32 { // this thread's byte of that query, 0x100 where it has none
        u32 const length = length_of(query);
        if (length == ~0u || position >= length) return 0x100u;
        return reinterpret_cast<u8 const *>(queries.base + query_offsets[query])[position];
    };
    u32 ahead_low = first_low, ahead_high = first_high;
#pragma unroll 1
    for (u32 group_first = 0; group_first < queries_here; group_first += tiny_group_k) {
        // masks: slot s lives in half s / 16 of dword s % 16 of every row; its query is right-aligned in the half's sixteen bits
        u32 const length_low = length_of(group_first + dword_of_mine), length_high = length_of(group_first + dword_of_mine + 16);
        if (ahead_low < 0x100u) atomicOr(&peq[ahead_low * tiny_row_dwords_k + dword_of_mine], 1u << (tiny_rows_k - length_low + position));
        if (ahead_high < 0x100u) atomicOr(&peq[ahead_high * tiny_row_dwords_k + dword_of_mine], 0x10000u << (tiny_rows_k - length_high + position));
        u32 const built_low = ahead_low, built_high = ahead_high;
        // the next group's bytes go out now and come back under the scoring below
        ahead_low = fetch(group_first + tiny_group_k + dword_of_mine), ahead_high = fetch(group_first + tiny_group_k + dword_of_mine + 16);
        // the group's 32 lengths: lane l of every wavefront works out slot l's, `readlane` hands them round as scalars
        u32 const length_of_my_slot = length_of(group_first + (lane & 31u));
        __syncthreads();
        if (group_first == 0) SZS_TINY_STAMP(3);
        // ---- sixteen registers of two patterns each; phantom low rows below a pattern shorter than 16
        u32 vp[16], vn[16];
        u32 skipped = 0; // bit s: slot s is a query this kernel leaves to the outliers' kernel (or lies past the span's end)
#pragma unroll
        for (u32 d = 0; d < 16; ++d) {
            u32 const low = __builtin_amdgcn_readlane(length_of_my_slot, d), high = __builtin_amdgcn_readlane(length_of_my_slot, d + 16);
            u32 const low_rows = low == ~0u ? 0u : low, high_rows = high == ~0u ? 0u : high;
            skipped |= (low == ~0u || group_first + d >= queries_here ? 1u : 0u) << d;
            skipped |= (high == ~0u || group_first + d + 16 >= queries_here ? 1u : 0u) << (d + 16);
            vp[d] = ((0xFFFFu << (tiny_rows_k - low_rows)) & 0xFFFFu) | ((0xFFFF0000u << (tiny_rows_k - high_rows)) & 0xFFFF0000u);
            vn[d] = 0;
        }
        uint4 const *const rows = reinterpret_cast<uint4 const *>(peq);
        auto take = [&](u32 symbol) {
            uint4 const *const row = rows + symbol * (tiny_row_dwords_k / 4);
            uint4 const a = row[0], b = row[1], c = row[2], e = row[3];
            u32 const masks[16] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w, e.x, e.y, e.z, e.w};
#pragma unroll
            for (u32 d = 0; d < 16; ++d) tiny_column(vp[d], vn[d], masks[d]);
        };
#pragma unroll
        for (u32 d = 0; d < 4; ++d) {
            if (4 * d >= longest_in_wave) break; // wave-uniform
#pragma unroll
            for (u32 step = 0; step < 4; ++step)
                if (4 * d + step < text_length) take((symbols[d] >> (8 * step)) & 0xFFu);
        }
        // ---- results: distance = text length + popcount(VP) - popcount(VN) per half (phantom rows hold zeros; at most 16, a byte
        //      each), through LDS - `out[j][column]` packs the slots j, j + 8, j + 16, j + 24 - so that every wavefront writes 512
        //      contiguous bytes of a row.  (Written straight from the registers - a lane its column, row by row - the launch took
        //      70 us instead of 67 with one group per workgroup and 115 instead of 77 with four: partial lines.)
#pragma unroll
        for (u32 j = 0; j < 8; ++j) {
            u32 packed = 0;
#pragma unroll
            for (u32 k = 0; k < 2; ++k) {
                u32 const d = j + 8 * k;
                u32 const low = text_length + (u32)__builtin_popcount(vp[d] & 0xFFFFu) - (u32)__builtin_popcount(vn[d] & 0xFFFFu);
                u32 const high = text_length + (u32)__builtin_popcount(vp[d] >> 16) - (u32)__builtin_popcount(vn[d] >> 16);
                packed |= (low << (8 * k)) | (high << (16 + 8 * k));
            }
            out[j * tiny_block_k + column] = packed;
        }
        if (group_first == 0) SZS_TINY_STAMP(4);
        __syncthreads();
        if (group_first == 0) SZS_TINY_STAMP(5);
        // un-build the masks (the same dwords back to zero: cheaper than clearing 16 KB) ...
        if (built_low < 0x100u) peq[built_low * tiny_row_dwords_k + dword_of_mine] = 0;
        if (built_high < 0x100u) peq[built_high * tiny_row_dwords_k + dword_of_mine] = 0;
        // ... and write the rows out: rows of skipped queries and columns of skipped candidates belong to the outliers' kernel
        if (my_tiny) {
            u64 *const first_row = results + (u64)(query_first + group_first) * results_row_stride + my_candidate;
            u32 packed[8];
#pragma unroll
            for (u32 j = 0; j < 8; ++j) packed[j] = out[j * tiny_block_k + tid];
#pragma unroll
            for (u32 s = 0; s < tiny_group_k; ++s)
                if (!((skipped >> s) & 1u)) first_row[(u64)s * results_row_stride] = (packed[s & 7u] >> (8 * (s >> 3))) & 0xFFu;
        }
        if (group_first == 0) SZS_TINY_STAMP(6);
        __syncthreads(); // the next group's atomics must not meet the un-building stores, nor its distances these reads
    }
    SZS_TINY_STAMP(7);
#undef SZS_TINY_STAMP
}


/* ---- round 5, second design: EVERY token in the one launch ---------------------------------------------------------------------
 *
 *  `levenshtein_tiny_whole_kernel` is the kernel above with the longer tokens (17 ... 255 bytes: a few per cent of a text's words)
 *  scored by the SAME workgroups, in the shadow of their tiny-token work, instead of by three launches around it (a pass that
 *  listed them and tabled the tiny strings' masks in device memory, the outliers' kernel, a pass that set the tables back: 6 + 35
 *  + 6 us beside the 50 of the tiny-token kernel, each bound by latency, not by work).  A workgroup meets three kinds of them:
 *
 *    A  a LONG CANDIDATE of its block (sorted behind the tiny ones by the local sort; its lane sits out the group's columns)
 *       against the group's thirty-two tiny queries: the text is the same for sixteen lanes, lane d of them advancing the two
 *       patterns of dword d of the group's masks - the very LDS rows the group's own columns read - one register each of VP /
 *       VN, a chain of eleven instructions per text byte instead of sixteen registers' worth.  Wavefront w takes the block's
 *       long candidates 4 w ... 4 w + 3 (then 16 further on): nobody waits for a wavefront that happens to hold them all.  The
 *       distances (at most 255: a byte) land in the group's `out` rows, and leave with the tiny ones' in whole runs.
 *    B  a LONG QUERY of its span against the block's tiny candidates: the query becomes an ordinary W-word Myers pattern (W = 1,
 *       2, 4, 8 by the span's longest) whose match masks are tabled in the LDS the groups' masks have left (16 KB: 16 / W
 *       patterns a round); every lane runs its own candidate - up to sixteen columns, the bytes still in its registers - over
 *       each of them.  Rows leave through `out` (16 bits a distance) as whole 2 KB runs.
 *    C  long query x long candidate: pattern r of a round is wavefront r % 4's, lane k streaming the block's k-th long candidate.
 *
 *  No list, no table in device memory, nothing to set back: ONE launch.  Malformed offsets or a string beyond 255 bytes leave
 *  `*unfit = unfit_sequence` (pinned memory) and the host scores the call the ordinary way.
 */
constexpr u32 tiny_longest_k = 255; // bytes of the longest string this kernel scores: a distance fits a byte of `out`

/**
 *  A long candidate's bytes, HELD by the sixteen lanes of a cluster: lane e of the cluster keeps the text's dwords e, 16 + e,
 *  32 + e and 48 + e (spliced to the text's own alignment) - four loads a lane, all in flight at once, ahead of the columns they
 *  feed.  The walks below then take the next four bytes from the cluster by `ds_bpermute`: no load from memory stands in a chain
 *  of dependent columns.  (Fetched inside the walk - a dword, then its four columns, then the next dword - every step waited a
 *  round trip to the L2: kinds A and C took 5 and 9 us a workgroup of 4096 x 4096 words, a quarter of the launch each.)
 */
struct tiny_held_text_t {
    u32 dwords[4];
    u32 column, length; // of the block; 0 bytes where the cluster has no text
};
__device__ __forceinline__ tiny_held_text_t tiny_hold_text(szs_tape_t const &candidates, u64 const *froms, u32 const *lengths, u32 const *lane_of_rank,
                                                            u32 tiny_count, u32 long_count, u32 k) {
    tiny_held_text_t held;
    bool const live = k < long_count;
    held.column = live ? lane_of_rank[tiny_count + k] : 0u;
    held.length = live ? lengths[held.column] : 0u;
    text_stream_t const text(candidates.base + froms[held.column], held.length);
    u32 const e = threadIdx.x & 15u;
    u32 raw[8];
#pragma unroll
    for (u32 c = 0; c < 4; ++c) raw[2 * c] = text.raw(16 * c + e), raw[2 * c + 1] = text.raw(16 * c + e + 1);
#pragma unroll
    for (u32 c = 0; c < 4; ++c) held.dwords[c] = text.splice(raw[2 * c], raw[2 * c + 1]);
    return held;
}
/** Bytes [4 i, 4 i + 4) of the text this lane's cluster holds (`i` the same for the whole wavefront). */
__device__ __forceinline__ u32 tiny_held_four(tiny_held_text_t const &held, u32 i) {
    u32 const chunk = i >> 4;
    u32 const mine = chunk == 0 ? held.dwords[0] : chunk == 1 ? held.dwords[1] : chunk == 2 ? held.dwords[2] : held.dwords[3];
    return (u32)__shfl((int)mine, (int)((threadIdx.x & 48u) + (i & 15u)), 64);
}

/** Kinds B and C: the span's long queries `listed[0 ... listed_count)` as W-word patterns, 16 / W of them a round. */
template <int words_>
__device__ __forceinline__ void tiny_long_queries(u32 *peq, u32 *out, u64 const *query_offsets, unsigned short const *listed, u32 listed_count,
                                                  szs_tape_t const &queries, szs_tape_t const &candidates, u32 query_first,
                                                  u64 const *froms, u32 const *lengths, u32 const *lane_of_rank, u32 tiny_count,
                                                  u32 long_count, tiny_held_text_t const &held_first, u32 column, bool column_is_tiny,
                                                  u32 text_length, u32 longest_in_wave, u32 const (&symbols)[4], bool my_exists,
                                                  u32 my_candidate, u64 *__restrict__ results, u64 results_row_stride) {
    constexpr u32 per_round = 16u / words_; // tables of 256 rows x W dwords in the 16 KB of `peq`
    constexpr u32 rows_k = 32u * words_;
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
        // ---- the round's tables: thread p ORs bit `pad + p` of pattern r into the row of the pattern's p-th byte
#pragma unroll 1
        for (u32 r = 0; r < here; ++r) {
            u32 const q = listed[first + r];
            u64 const from = query_offsets[q];
            u32 const length = (u32)(query_offsets[q + 1] - from);
            if (tid < length) {
                u32 const byte = reinterpret_cast<u8 const *>(queries.base + from)[tid], bit = rows_k - length + tid;
                atomicOr(&peq[(r * 256u + byte) * words_ + (bit >> 5)], 1u << (bit & 31u));
            }
        }
        __syncthreads();
        // ---- B: this lane's tiny candidate under every pattern of the round
#pragma unroll 1
        for (u32 r = 0; r < here; ++r) {
            u32 const q = listed[first + r];
            u32 const length = (u32)(query_offsets[q + 1] - query_offsets[q]);
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
        // ---- C: the block's long candidates, a cluster of sixteen lanes each; lane r of a cluster takes pattern r of the round
#pragma unroll 1
        for (u32 k_first = 0; k_first < long_count; k_first += 16) {
            tiny_held_text_t const held = k_first ? tiny_hold_text(candidates, froms, lengths, lane_of_rank, tiny_count, long_count, k_first + wave * 4 + (lane >> 4)) : held_first;
            u32 const longest = (u32)__builtin_amdgcn_readfirstlane(wave_max_u32(held.length));
            u32 const r = lane & 15u;
            bool const live = r < here && held.length;
            u32 const q = listed[first + (r < here ? r : 0u)];
            u32 const length = (u32)(query_offsets[q + 1] - query_offsets[q]);
            u32 vp[words_], vn[words_];
            start(rows_k - length, vp, vn);
            u32 const table = r < here ? r : 0u;
            if constexpr (words_ <= 2) { // the next step's masks in flight under this step's columns
                u32 eq_now[4][words_], eq_next[4][words_];
                u32 four_next = tiny_held_four(held, 0);
#pragma unroll
                for (u32 i = 0; i < 4; ++i) masks_of(table, (four_next >> (8 * i)) & 0xFFu, eq_now[i]);
                four_next = tiny_held_four(held, 1);
#pragma unroll 1
                for (u32 at = 0; at < longest; at += 4) {
#pragma unroll
                    for (u32 i = 0; i < 4; ++i) masks_of(table, (four_next >> (8 * i)) & 0xFFu, eq_next[i]);
                    four_next = tiny_held_four(held, at / 4 + 2);
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
                u32 four_now = tiny_held_four(held, 0);
#pragma unroll 1
                for (u32 at = 0; at < longest; at += 4) {
                    u32 const four_next = tiny_held_four(held, at / 4 + 1);
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

__global__ __launch_bounds__(256, 4) void levenshtein_tiny_whole_kernel(szs_tape_t queries, szs_tape_t candidates, u32 queries_per_workgroup,
                                                                    u64 *__restrict__ results, u64 results_row_stride, u32 *unfit,
                                                                    u32 unfit_sequence, unsigned long long *symbols_out, u64 *trace, u32 debug_skip) {
#define SZS_TINY_STAMP(K) do { if (trace && threadIdx.x == 0) trace[(u64)blockIdx.x * 10 + (K)] = wall_clock64(); } while (0)
    SZS_TINY_STAMP(0);
    __shared__ __attribute__((aligned(16))) u32 peq[256 * tiny_row_dwords_k]; // the group's masks; afterwards the long queries' tables: 16 KB
    __shared__ __attribute__((aligned(16))) u32 out[8 * tiny_block_k];        // [j][column of the block], a byte a distance: 8 KB
    __shared__ u64 query_offsets[tiny_most_queries_k + 1];
    __shared__ u64 froms[tiny_block_k];
    __shared__ u32 lengths[tiny_block_k], bins[32], lane_of_rank[tiny_block_k];
    __shared__ unsigned short listed[tiny_most_queries_k]; // the span's long queries (17 ... 255 bytes), by their place in the span
    __shared__ u32 listed_count, listed_longest;

    u32 const tid = threadIdx.x;
    u32 const blocks = (candidates.count + tiny_block_k - 1) / tiny_block_k;
    u32 const block = blockIdx.x % blocks, span = blockIdx.x / blocks;
    u32 const query_first = span * queries_per_workgroup;
    u32 const queries_here = queries.count - query_first < queries_per_workgroup ? queries.count - query_first : queries_per_workgroup;
    if (blockIdx.x == 0 && tid < 2 && symbols_out) { // the call's cell count is the product of these two (the host's profile)
        szs_tape_t const &whole = tid ? candidates : queries;
        symbols_out[tid] = tiny_offset(whole.offsets, whole.wide, whole.count) - tiny_offset(whole.offsets, whole.wide, 0);
    }
    // ---- the FIRST group's query bytes: requested before anything else, straight from the tape's offsets
    u32 first_low = 0x100u, first_high = 0x100u;
    {
        u32 const at = tid & 15u;
        u32 const slots[2] = {tid >> 4, (tid >> 4) + 16};
        u64 from[2] = {0, 0}, to[2] = {0, 0};
#pragma unroll
        for (u32 k = 0; k < 2; ++k)
            if (slots[k] < queries_here)
                from[k] = tiny_offset(queries.offsets, queries.wide, (u64)query_first + slots[k]), to[k] = tiny_offset(queries.offsets, queries.wide, (u64)query_first + slots[k] + 1);
        if (to[0] >= from[0] && to[0] - from[0] <= tiny_rows_k && at < to[0] - from[0]) first_low = reinterpret_cast<u8 const *>(queries.base + from[0])[at];
        if (to[1] >= from[1] && to[1] - from[1] <= tiny_rows_k && at < to[1] - from[1]) first_high = reinterpret_cast<u8 const *>(queries.base + from[1])[at];
    }
    // ---- once per workgroup: the block's candidates (offsets, local sort by length: tiny ones, then the long ones, then the absent)
    u32 const my_candidate = block * tiny_block_k + tid;
    u64 my_from = 0;
    u32 my_length = 0;
    bool my_exists = false; // this thread's candidate exists and this kernel scores it (up to 255 bytes)
    if (my_candidate < candidates.count) {
        my_from = tiny_offset(candidates.offsets, candidates.wide, my_candidate);
        u64 const to = tiny_offset(candidates.offsets, candidates.wide, (u64)my_candidate + 1);
        if (to >= my_from && to - my_from <= tiny_longest_k) my_length = (u32)(to - my_from), my_exists = true;
        else *unfit = unfit_sequence; // malformed, or too long for this kernel: the host scores the call the ordinary way
    }
    for (u32 i = tid; i <= queries_here; i += 256) query_offsets[i] = tiny_offset(queries.offsets, queries.wide, (u64)query_first + i);
    for (u32 i = tid; i < 256 * tiny_row_dwords_k; i += 256) peq[i] = 0;
    if (tid < 32) bins[tid] = 0;
    if (tid == 0) listed_count = 0, listed_longest = 0;
    froms[tid] = my_from, lengths[tid] = my_exists ? my_length : 0x80000000u;
    __syncthreads();
    u32 const bin = !my_exists ? tiny_rows_k + 2 : my_length <= tiny_rows_k ? my_length : tiny_rows_k + 1;
    u32 const place_in_bin = atomicAdd(&bins[bin], 1u);
    for (u32 i = tid; i < queries_here; i += 256) { // the span's long queries, in whatever order the atomics hand out
        u64 const from = query_offsets[i], to = query_offsets[i + 1];
        if (to < from || to - from > tiny_longest_k) *unfit = unfit_sequence;
        else if (to - from > tiny_rows_k) listed[atomicAdd(&listed_count, 1u)] = (unsigned short)i, atomicMax(&listed_longest, (u32)(to - from));
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
    __syncthreads();
    SZS_TINY_STAMP(1);
    u32 const tiny_count = bins[tiny_rows_k + 1], long_count = bins[tiny_rows_k + 2] - tiny_count; // of the block's candidates
    // this lane SCORES the candidate of rank ((wave + block) % 4) x 64 + lane - column `column` of the block
    u32 const column = lane_of_rank[(((tid >> 6) + blockIdx.x) & 3u) * 64u + (tid & 63u)];
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
    u32 const dword_of_mine = tid >> 4, position = tid & 15u, lane = tid & 63u, wave = tid >> 6;
    // the block's first sixteen long candidates, four to a wavefront, held by clusters of sixteen lanes for kinds A and C
    tiny_held_text_t const held_first = tiny_hold_text(candidates, froms, lengths, lane_of_rank, tiny_count, long_count, wave * 4 + (lane >> 4));
    SZS_TINY_STAMP(2);

    auto length_of = [&](u32 query) -> u32 { // of a query of the span; 0 past the span's end, ~0 for one the groups skip
        if (query >= queries_here) return 0;
        u64 const from = query_offsets[query], to = query_offsets[query + 1];
        return to >= from && to - from <= tiny_rows_k ? (u32)(to - from) : ~0u;
    };
    auto fetch = [&](u32 query) -> u32 { // this thread's byte of that query, 0x100 where it has none
        u32 const length = length_of(query);
        if (length == ~0u || position >= length) return 0x100u;
        return reinterpret_cast<u8 const *>(queries.base + query_offsets[query])[position];
    };
    u32 ahead_low = first_low, ahead_high = first_high;
#pragma unroll 1
    for (u32 group_first = 0; group_first < queries_here; group_first += tiny_group_k) {
        u32 const length_low = length_of(group_first + dword_of_mine), length_high = length_of(group_first + dword_of_mine + 16);
        if (ahead_low < 0x100u) atomicOr(&peq[ahead_low * tiny_row_dwords_k + dword_of_mine], 1u << (tiny_rows_k - length_low + position));
        if (ahead_high < 0x100u) atomicOr(&peq[ahead_high * tiny_row_dwords_k + dword_of_mine], 0x10000u << (tiny_rows_k - length_high + position));
        u32 const built_low = ahead_low, built_high = ahead_high;
        ahead_low = fetch(group_first + tiny_group_k + dword_of_mine), ahead_high = fetch(group_first + tiny_group_k + dword_of_mine + 16);
        u32 const length_of_my_slot = length_of(group_first + (lane & 31u));
        __syncthreads();
        if (group_first == 0) SZS_TINY_STAMP(3);
        u32 skipped = 0; // bit s: slot s is a long query (kind B's) or lies past the span's end
        {
            u32 vp[16], vn[16];
#pragma unroll
            for (u32 d = 0; d < 16; ++d) {
                u32 const low = __builtin_amdgcn_readlane(length_of_my_slot, d), high = __builtin_amdgcn_readlane(length_of_my_slot, d + 16);
                u32 const low_rows = low == ~0u ? 0u : low, high_rows = high == ~0u ? 0u : high;
                skipped |= (low == ~0u || group_first + d >= queries_here ? 1u : 0u) << d;
                skipped |= (high == ~0u || group_first + d + 16 >= queries_here ? 1u : 0u) << (d + 16);
                vp[d] = ((0xFFFFu << (tiny_rows_k - low_rows)) & 0xFFFFu) | ((0xFFFF0000u << (tiny_rows_k - high_rows)) & 0xFFFF0000u);
                vn[d] = 0;
            }
            uint4 const *const rows = reinterpret_cast<uint4 const *>(peq);
            auto take = [&](u32 symbol) {
                uint4 const *const row = rows + symbol * (tiny_row_dwords_k / 4);
                uint4 const a = row[0], b = row[1], c = row[2], e = row[3];
                u32 const masks[16] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w, e.x, e.y, e.z, e.w};
#pragma unroll
                for (u32 d = 0; d < 16; ++d) tiny_column(vp[d], vn[d], masks[d]);
            };
#pragma unroll
            for (u32 d = 0; d < 4; ++d) {
                if (4 * d >= longest_in_wave) break; // wave-uniform
#pragma unroll
                for (u32 at = 0; at < 4; ++at)
                    if (4 * d + at < text_length) take((symbols[d] >> (8 * at)) & 0xFFu);
            }
            if (column_is_tiny) { // (a long column's bytes of `out` are kind A's to write)
#pragma unroll
                for (u32 j = 0; j < 8; ++j) {
                    u32 packed = 0;
#pragma unroll
                    for (u32 k = 0; k < 2; ++k) {
                        u32 const d = j + 8 * k;
                        u32 const low = text_length + (u32)__builtin_popcount(vp[d] & 0xFFFFu) - (u32)__builtin_popcount(vn[d] & 0xFFFFu);
                        u32 const high = text_length + (u32)__builtin_popcount(vp[d] >> 16) - (u32)__builtin_popcount(vn[d] >> 16);
                        packed |= (low << (8 * k)) | (high << (16 + 8 * k));
                    }
                    out[j * tiny_block_k + column] = packed;
                }
            }
        }
        if (group_first == 0) SZS_TINY_STAMP(4);
        // ---- A: the block's long candidates under the group's masks - sixteen lanes a text, lane d of them the patterns of dword d
#pragma unroll 1
        for (u32 k_first = 0; k_first < long_count && !(debug_skip & 1u); k_first += 16) {
            tiny_held_text_t const held = k_first ? tiny_hold_text(candidates, froms, lengths, lane_of_rank, tiny_count, long_count, k_first + wave * 4 + (lane >> 4)) : held_first;
            u32 const d = lane & 15u;
            u32 const longest = (u32)__builtin_amdgcn_readfirstlane(wave_max_u32(held.length));
            u32 const low = length_of(group_first + d), high = length_of(group_first + d + 16);
            u32 const low_rows = low == ~0u ? 0u : low, high_rows = high == ~0u ? 0u : high;
            u32 vp = ((0xFFFFu << (tiny_rows_k - low_rows)) & 0xFFFFu) | ((0xFFFF0000u << (tiny_rows_k - high_rows)) & 0xFFFF0000u), vn = 0;
            u32 masks_now[4], masks_next[4];
            u32 four_next = tiny_held_four(held, 0);
#pragma unroll
            for (u32 i = 0; i < 4; ++i) masks_now[i] = peq[((four_next >> (8 * i)) & 0xFFu) * tiny_row_dwords_k + d];
            four_next = tiny_held_four(held, 1);
#pragma unroll 1
            for (u32 at = 0; at < longest; at += 4) {
#pragma unroll
                for (u32 i = 0; i < 4; ++i) masks_next[i] = peq[((four_next >> (8 * i)) & 0xFFu) * tiny_row_dwords_k + d]; // the next step's masks ...
                four_next = tiny_held_four(held, at / 4 + 2);                                                               // ... and the bytes of the one after
#pragma unroll
                for (u32 i = 0; i < 4; ++i)
                    if (at + i < held.length) tiny_column(vp, vn, masks_now[i]);
#pragma unroll
                for (u32 i = 0; i < 4; ++i) masks_now[i] = masks_next[i];
            }
            if (held.length) {
                u8 *const bytes = reinterpret_cast<u8 *>(out) + ((u64)((d & 7u) * tiny_block_k + held.column)) * 4u + (d >> 3);
                bytes[0] = (u8)(held.length + (u32)__builtin_popcount(vp & 0xFFFFu) - (u32)__builtin_popcount(vn & 0xFFFFu));
                bytes[2] = (u8)(held.length + (u32)__builtin_popcount(vp >> 16) - (u32)__builtin_popcount(vn >> 16));
            }
        }
        __syncthreads();
        if (group_first == 0) SZS_TINY_STAMP(5);
        // un-build the masks (the same dwords back to zero: cheaper than clearing 16 KB) ...
        if (built_low < 0x100u) peq[built_low * tiny_row_dwords_k + dword_of_mine] = 0;
        if (built_high < 0x100u) peq[built_high * tiny_row_dwords_k + dword_of_mine] = 0;
        // ... and write the rows out, whole runs: the rows of long queries are kind B's
        if (my_exists) {
            u64 *const first_row = results + (u64)(query_first + group_first) * results_row_stride + my_candidate;
            u32 packed[8];
#pragma unroll
            for (u32 j = 0; j < 8; ++j) packed[j] = out[j * tiny_block_k + tid];
#pragma unroll
            for (u32 s = 0; s < tiny_group_k; ++s)
                if (!((skipped >> s) & 1u)) first_row[(u64)s * results_row_stride] = (packed[s & 7u] >> (8 * (s >> 3))) & 0xFFu;
        }
        if (group_first == 0) SZS_TINY_STAMP(6);
        __syncthreads(); // the next group's atomics must not meet the un-building stores, nor its distances these reads
    }
    SZS_TINY_STAMP(7);
    // ---- B and C: the span's long queries, as W-word patterns by the longest of them
    u32 const long_queries = listed_count;
    if (long_queries && !(debug_skip & 2u)) {
        u32 const longest_query = listed_longest;
#define SZS_TINY_LONG(W)                                                                                                                   \
    tiny_long_queries<W>(peq, out, query_offsets, listed, long_queries, queries, candidates, query_first, froms, lengths, lane_of_rank,   \
                         tiny_count, long_count, held_first, column, column_is_tiny, text_length, longest_in_wave, symbols, my_exists,    \
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

} // namespace szs_hip

extern "C" int szs_hip_levenshtein_tiny_prepare(szs_tape_t const *queries, szs_tape_t const *candidates, uint32_t *unfit, uint32_t unfit_sequence,
                                                szs_tiny_outliers_t *outliers, uint32_t *query_masks, uint32_t *candidate_masks,
                                                unsigned long long *symbols_out, int unbuild, void *stream) {
    using namespace szs_hip;
    if (!queries->count || !candidates->count) return 0;
    u64 const query_workgroups = ((u64)queries->count + 255) / 256, candidate_workgroups = ((u64)candidates->count + 255) / 256;
    hipLaunchKernelGGL(levenshtein_tiny_prepare_kernel, dim3((u32)(query_workgroups + candidate_workgroups)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), *queries, *candidates, (u32)query_workgroups, unfit, unfit_sequence, outliers, query_masks,
                       candidate_masks, symbols_out, unbuild);
    return (int)hipGetLastError();
}

extern "C" int szs_hip_levenshtein_tiny(szs_tape_t const *queries_tape, szs_tape_t const *candidates_tape, uint64_t *results,
                                        uint64_t results_row_stride, uint64_t *trace, void *stream) {
    using namespace szs_hip;
    szs_tape_t const queries = *queries_tape, candidates = *candidates_tape;
    u32 const queries_count = queries.count, candidates_count = candidates.count;
    if (!queries_count || !candidates_count) return 0;
    u64 const blocks = ((u64)candidates_count + tiny_block_k - 1) / tiny_block_k;
    // Spans of the queries: enough workgroups to fill the device a few times over (a workgroup's set-up - offsets, the local sort -
    // is paid once per span), whole groups of thirty-two, at most tiny_most_queries_k queries each.
    // (measured on 4096 x 4096 words: 2048 workgroups of one group each 55.6 us, 1024 of two 50.5, 688 of three 63.0 - a workgroup's
    // set-up is 7 of its 20 us, but fewer workgroups than 4 per CU leave nobody to run while the others wait)
    u64 const wanted_workgroups = 1024;
    u64 spans = (wanted_workgroups + blocks - 1) / blocks;
    u64 per_span = ((u64)queries_count + spans - 1) / spans;
    per_span = (per_span + tiny_group_k - 1) / tiny_group_k * tiny_group_k;
    if (per_span > tiny_most_queries_k) per_span = tiny_most_queries_k;
    spans = ((u64)queries_count + per_span - 1) / per_span;
    if (blocks * spans > 0x7FFFFFFFull) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(levenshtein_tiny_kernel, dim3((u32)(blocks * spans)), dim3(256), 0, static_cast<hipStream_t>(stream), queries, candidates,
                       (u32)per_span, results, results_row_stride, trace);
    return (int)hipGetLastError();
}

extern "C" int szs_hip_levenshtein_tiny_whole(szs_tape_t const *queries_tape, szs_tape_t const *candidates_tape, uint64_t *results,
                                              uint64_t results_row_stride, uint32_t *unfit, uint32_t unfit_sequence,
                                              unsigned long long *symbols_out, uint64_t *trace, void *stream) {
    using namespace szs_hip;
    szs_tape_t const queries = *queries_tape, candidates = *candidates_tape;
    u32 const queries_count = queries.count, candidates_count = candidates.count;
    if (!queries_count || !candidates_count) return 0;
    static int debug_skip = -1;
    if (debug_skip < 0) debug_skip = getenv("SZS_DEBUG_SKIP") ? atoi(getenv("SZS_DEBUG_SKIP")) : 0;
    u64 const blocks = ((u64)candidates_count + tiny_block_k - 1) / tiny_block_k;
    u64 const wanted_workgroups = 1024; // (as above: two groups a workgroup on 4096 x 4096 words)
    u64 spans = (wanted_workgroups + blocks - 1) / blocks;
    u64 per_span = ((u64)queries_count + spans - 1) / spans;
    per_span = (per_span + tiny_group_k - 1) / tiny_group_k * tiny_group_k;
    if (per_span > tiny_most_queries_k) per_span = tiny_most_queries_k;
    spans = ((u64)queries_count + per_span - 1) / per_span;
    if (blocks * spans > 0x7FFFFFFFull) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(levenshtein_tiny_whole_kernel, dim3((u32)(blocks * spans)), dim3(256), 0, static_cast<hipStream_t>(stream), queries,
                       candidates, (u32)per_span, results, results_row_stride, unfit, unfit_sequence, symbols_out, trace, (u32)debug_skip);
    return (int)hipGetLastError();
}
