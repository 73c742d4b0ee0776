/*
 *  planner.hip - the O(Q + C) planner on the DEVICE: tape offsets in, length-sorted string refs out, no host round trip.
 *
 *  The reference plans per CELL on the device - a 112-byte task per (query, candidate) pair, counting-sorted by size tier
 *  and scattered afterwards (/root/reference/include/stringzillas/similarities/cuda.cuh:1652-1711,1887-1957) - or, on its
 *  fast path, skips task materialisation altogether (cuda.cuh:4297-4340).  This build plans per ROW and COLUMN
 *  (host/plan.c); round 1 did that on the host, which cost a download of the offsets and a stream synchronisation before
 *  the first scoring launch could even be enqueued (18 % of config 2's wall time).  Here the same plan is produced by ONE
 *  workgroup of 1024 threads straight from the caller's offsets:
 *
 *    pass 1  per side: lengths, their maximum and sum, the band counts the tier model wants, the number of strings per
 *            bit-parallel launch variant; offsets that descend or strings of 4 GiB are flagged, not scored;
 *    check   the host may have enqueued the scoring launches ALREADY, shaped like the previous call of this engine (same
 *            launch variants, workspaces sized for the previous longest strings).  If this batch does not fit that shape,
 *            every ref is written with length 0 - the speculated launches then score empty strings, memory-safe and over
 *            in microseconds - and `speculation_held` stays 0; the host re-plans from the summary and launches again;
 *    pass 2  per side: counting sort by length in LDS (histogram, block-wide exclusive scan, scatter) into TWO ref arrays,
 *            ascending (the lane side of every kernel) and descending (the workgroup side: longest first makes every
 *            launch variant a contiguous slice and hands out the heaviest workgroups first).
 *
 *  The summary lands in pinned host memory; the host reads it after the call's ONE synchronisation.
 *  Strings of `plan_bins_k` bytes or more are not sorted here (`status` says so; the host planner takes over).
 */
#include "device_common.hpp"

namespace szs_hip {

constexpr int plan_threads_k = 1024;
constexpr u32 plan_bins_k = SZS_PLAN_DEVICE_LONGEST + 1; // lengths below this are counting-sorted in LDS

__device__ __forceinline__ u64 tape_offset(void const *offsets, u32 wide, u64 index) {
    return wide ? static_cast<u64 const *>(offsets)[index] : (u64) static_cast<u32 const *>(offsets)[index];
}

/** Index into `variant_counts` of a string of `length` symbols: 0 = no bit-parallel width takes it. */
__device__ __forceinline__ u32 variant_slot(u32 length, u32 myers_words) {
    if (!myers_words) return 0;
    u32 const words = length ? (length + 31u) / 32u : 1u;
    if (words > myers_words) return 0;
    // SZS_MYERS_SHORT_WORDS, 10, 12, 16, 20, 24, 32, 48, 64 - szs_hip_levenshtein_myers_round_words()
    return words <= 8 ? 1 : words <= 10 ? 2 : words <= 12 ? 3 : words <= 16 ? 4 : words <= 20 ? 5 : words <= 24 ? 6 : words <= 32 ? 7 : words <= 48 ? 8 : 9;
}

/** Block-wide exclusive scan of one value per thread (1024 threads = 16 wavefronts); returns this thread's prefix and
 *  leaves the block total in `*total`.  `scratch` holds 17 words. */
__device__ __forceinline__ u32 block_exclusive_scan(u32 value, u32 *scratch, u32 *total) {
    u32 const lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    u32 inclusive = value;
#pragma unroll
    for (int offset = 1; offset < 64; offset <<= 1) {
        u32 const other = (u32)__shfl_up((int)inclusive, offset, 64);
        if (lane >= (u32)offset) inclusive += other;
    }
    __syncthreads(); // scratch may still be read by a previous call
    if (lane == 63) scratch[wave] = inclusive;
    __syncthreads();
    if (threadIdx.x == 0) {
        u32 running = 0;
        for (int w = 0; w < plan_threads_k / 64; ++w) {
            u32 const sum = scratch[w];
            scratch[w] = running, running += sum;
        }
        scratch[plan_threads_k / 64] = running;
    }
    __syncthreads();
    *total = scratch[plan_threads_k / 64];
    return scratch[wave] + inclusive - value;
}

__global__ __launch_bounds__(plan_threads_k) void plan_kernel(szs_plan_side_t queries, szs_plan_side_t candidates,
                                                              int symmetric, u32 myers_words,
                                                              szs_plan_expectation_t expected,
                                                              szs_plan_summary_t *__restrict__ summary) {
    __shared__ u32 histogram[plan_bins_k];
    __shared__ u32 scan_scratch[plan_threads_k / 64 + 1];
    __shared__ u32 shared_longest[2], shared_status, shared_variants[2][SZS_PLAN_VARIANTS];
    __shared__ unsigned long long shared_symbols[2], shared_bands_systolic[2], shared_bands_chain[2], shared_cells;
    __shared__ u32 shared_held;

    u32 const tid = threadIdx.x;
    if (tid < 2) shared_longest[tid] = 0, shared_symbols[tid] = 0, shared_bands_systolic[tid] = 0, shared_bands_chain[tid] = 0;
    if (tid < 2 * SZS_PLAN_VARIANTS) shared_variants[tid / SZS_PLAN_VARIANTS][tid % SZS_PLAN_VARIANTS] = 0;
    if (tid == 0) shared_status = 0, shared_cells = 0, shared_held = 0;
    __syncthreads();

    int const sides = symmetric ? 1 : 2;
    // ---- pass 1: lengths and the statistics every decision is made from
    for (int s = 0; s < sides; ++s) {
        szs_plan_side_t const &side = s ? candidates : queries;
        u32 longest = 0, status = 0, variants[SZS_PLAN_VARIANTS] = {0};
        u64 symbols = 0, bands_systolic = 0, bands_chain = 0;
        for (u32 i = tid; i < side.count; i += plan_threads_k) {
            u64 const from = tape_offset(side.offsets, side.wide, i), to = tape_offset(side.offsets, side.wide, (u64)i + 1);
            if (to < from) status |= SZS_PLAN_STATUS_DESCENDING;
            u64 const wide_length = to < from ? 0 : to - from;
            if (wide_length > 0xFFFFFFFFull) status |= SZS_PLAN_STATUS_OVERFLOW;
            u32 const length = (u32)wide_length;
            longest = length > longest ? length : longest;
            symbols += length;
            bands_systolic += length ? (length + SZS_SYSTOLIC_BAND_ROWS - 1) / SZS_SYSTOLIC_BAND_ROWS : 1;
            bands_chain += length ? (length + SZS_MYERS_CHAIN_BAND_ROWS - 1) / SZS_MYERS_CHAIN_BAND_ROWS : 1;
            u32 const slot = variant_slot(length, myers_words);
#pragma unroll
            for (u32 v = 0; v < SZS_PLAN_VARIANTS; ++v) variants[v] += slot == v;
        }
        atomicMax(&shared_longest[s], longest);
        if (status) atomicOr(&shared_status, status);
        atomicAdd(&shared_symbols[s], (unsigned long long)symbols);
        atomicAdd(&shared_bands_systolic[s], (unsigned long long)bands_systolic);
        atomicAdd(&shared_bands_chain[s], (unsigned long long)bands_chain);
#pragma unroll
        for (u32 v = 0; v < SZS_PLAN_VARIANTS; ++v)
            if (variants[v]) atomicAdd(&shared_variants[s][v], variants[v]);
    }
    __syncthreads();

    // ---- symmetric calls: cells of the lower triangle = sum_i len_i * sum_{j <= i} len_j, in the caller's order
    if (symmetric) {
        u32 const chunk = (queries.count + plan_threads_k - 1) / plan_threads_k;
        u32 const first = tid * chunk, last = first + chunk < queries.count ? first + chunk : queries.count;
        u64 mine = 0;
        for (u32 i = first; i < last; ++i)
            mine += tape_offset(queries.offsets, queries.wide, (u64)i + 1) - tape_offset(queries.offsets, queries.wide, i);
        // prefix of the chunk sums: 64-bit, so two 32-bit scans would not do - a serial pass by one thread is 1024 adds
        __shared__ unsigned long long chunk_sums[plan_threads_k];
        chunk_sums[tid] = mine;
        __syncthreads();
        if (tid == 0) {
            unsigned long long running = 0;
            for (int t = 0; t < plan_threads_k; ++t) {
                unsigned long long const sum = chunk_sums[t];
                chunk_sums[t] = running, running += sum;
            }
        }
        __syncthreads();
        u64 running = chunk_sums[tid], cells = 0;
        for (u32 i = first; i < last; ++i) {
            u64 const length = tape_offset(queries.offsets, queries.wide, (u64)i + 1) - tape_offset(queries.offsets, queries.wide, i);
            running += length, cells += length * running;
        }
        atomicAdd(&shared_cells, (unsigned long long)cells);
        __syncthreads();
    }

    // ---- does this batch have the shape the host already enqueued launches for?
    if (tid == 0) {
        u32 held = expected.enabled && !shared_status;
        if (held) {
            int const query_side = symmetric ? 0 : (int)expected.query_side;
            for (u32 v = 0; v < SZS_PLAN_VARIANTS; ++v) held &= shared_variants[query_side][v] == expected.variant_counts[v];
            held &= shared_longest[0] <= expected.longest[0];
            held &= shared_longest[symmetric ? 0 : 1] <= expected.longest[1];
        }
        shared_held = held;
    }
    __syncthreads();
    bool const blank = expected.enabled && !shared_held; // speculated launches must find nothing to score
    u32 unsorted = 0;

    // ---- pass 2: counting sort by length, ascending and descending ref arrays
    for (int s = 0; s < sides; ++s) {
        szs_plan_side_t const &side = s ? candidates : queries;
        u32 const longest = shared_longest[s];
        if (shared_status || longest >= plan_bins_k) { // the host planner takes over
            unsorted = 1;
            continue;
        }
        u32 const bins = longest + 1;
        __syncthreads(); // the histogram of the previous side is done with
        for (u32 b = tid; b < bins; b += plan_threads_k) histogram[b] = 0;
        __syncthreads();
        for (u32 i = tid; i < side.count; i += plan_threads_k)
            atomicAdd(&histogram[(u32)(tape_offset(side.offsets, side.wide, (u64)i + 1) - tape_offset(side.offsets, side.wide, i))], 1u);
        __syncthreads();
        u32 const chunk = (bins + plan_threads_k - 1) / plan_threads_k;
        u32 const first_bin = tid * chunk, last_bin = first_bin + chunk < bins ? first_bin + chunk : bins;
        u32 mine = 0;
        for (u32 b = first_bin; b < last_bin; ++b) mine += histogram[b];
        u32 total;
        u32 running = block_exclusive_scan(mine, scan_scratch, &total);
        for (u32 b = first_bin; b < last_bin; ++b) {
            u32 const here = histogram[b];
            histogram[b] = running, running += here;
        }
        __syncthreads();
        for (u32 i = tid; i < side.count; i += plan_threads_k) {
            u64 const from = tape_offset(side.offsets, side.wide, i);
            u32 const length = (u32)(tape_offset(side.offsets, side.wide, (u64)i + 1) - from);
            u32 const position = atomicAdd(&histogram[length], 1u); // equal lengths: any order scores the same matrix
            szs_string_ref_t ref;
            ref.address = side.base + from, ref.length = blank ? 0u : length, ref.index = i;
            side.ascending[position] = ref;
            side.descending[side.count - 1 - position] = ref;
        }
    }
    __syncthreads();

    if (tid == 0) {
        summary->status = shared_status | (unsorted ? SZS_PLAN_STATUS_UNSORTED : 0u);
        summary->speculation_held = shared_held;
        for (int s = 0; s < 2; ++s) {
            int const from = symmetric ? 0 : s;
            summary->side[s].count = from ? candidates.count : queries.count;
            summary->side[s].longest = shared_longest[from];
            summary->side[s].symbols = shared_symbols[from];
            summary->side[s].bands_systolic = shared_bands_systolic[from];
            summary->side[s].bands_chain = shared_bands_chain[from];
            for (u32 v = 0; v < SZS_PLAN_VARIANTS; ++v) summary->variant_counts[s][v] = shared_variants[from][v];
        }
        summary->symmetric_cells = shared_cells;
        __threadfence_system();
        summary->sequence = expected.sequence; // written last: the host can tell a fresh summary from a stale one
    }
}

} // namespace szs_hip

extern "C" int szs_hip_plan(szs_plan_side_t const *queries, szs_plan_side_t const *candidates, unsigned myers_words,
                            szs_plan_expectation_t const *expected, szs_plan_summary_t *summary, void *stream) {
    using namespace szs_hip;
    szs_plan_expectation_t none = {};
    hipLaunchKernelGGL(plan_kernel, dim3(1), dim3(plan_threads_k), 0, static_cast<hipStream_t>(stream), *queries,
                       candidates ? *candidates : *queries, candidates ? 0 : 1, (u32)myers_words, expected ? *expected : none,
                       summary);
    return (int)hipGetLastError();
}
