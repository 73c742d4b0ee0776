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

/** Statistics of one side, reduced over the workgroup WITHOUT atomics: hipcc turns an LDS atomic on a wave-uniform address
 *  into a scalar loop over the 64 lanes (7 SALU instructions per lane and atomic - fifteen of them made the planner a 40 us
 *  kernel); a shuffle butterfly per value and one LDS slot per wavefront is ~300 instructions. */
constexpr int plan_values_k = 5 + SZS_PLAN_VARIANTS; // longest (max), status (or), symbols, bands x 2, strings per variant (sums)

__device__ __forceinline__ u64 shuffle_xor_u64(u64 value, int offset) {
    u32 const low = (u32)__shfl_xor((int)(u32)value, offset, 64), high = (u32)__shfl_xor((int)(u32)(value >> 32), offset, 64);
    return ((u64)high << 32) | low;
}

__global__ __launch_bounds__(plan_threads_k) void plan_kernel(szs_plan_side_t queries, szs_plan_side_t candidates,
                                                              int symmetric, u32 myers_words,
                                                              szs_plan_expectation_t expected,
                                                              szs_plan_summary_t *__restrict__ summary) {
    __shared__ u32 histogram[plan_bins_k];
    __shared__ u32 scan_scratch[plan_threads_k / 64 + 1];
    __shared__ unsigned long long wave_values[2][plan_threads_k / 64][plan_values_k];
    __shared__ unsigned long long side_values[2][plan_values_k];
    __shared__ unsigned long long chunk_sums[plan_threads_k], shared_cells;
    __shared__ u32 shared_held;

    u32 const tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    int const sides = symmetric ? 1 : 2;
    // The planner is a chain of dependent memory round trips, not of instructions: a load from HBM and back is a microsecond
    // or two.  So every thread fetches the offsets of ITS first string of both sides up front - one round trip for the whole
    // kernel when a side has at most 1024 strings - and all three passes read them from registers.
    u64 first_from[2] = {0, 0}, first_to[2] = {0, 0};
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        szs_plan_side_t const &side = s ? candidates : queries;
        if (s < sides && tid < side.count)
            first_from[s] = tape_offset(side.offsets, side.wide, tid), first_to[s] = tape_offset(side.offsets, side.wide, (u64)tid + 1);
    }
    auto span_of = [&](int s, szs_plan_side_t const &side, u32 i, u64 &from, u64 &to) {
        if (i == tid) from = first_from[s], to = first_to[s];
        else from = tape_offset(side.offsets, side.wide, i), to = tape_offset(side.offsets, side.wide, (u64)i + 1);
    };
    // ---- pass 1: lengths and the statistics every decision is made from
    for (int s = 0; s < sides; ++s) {
        szs_plan_side_t const &side = s ? candidates : queries;
        u64 values[plan_values_k];
#pragma unroll
        for (int k = 0; k < plan_values_k; ++k) values[k] = 0;
        for (u32 i = tid; i < side.count; i += plan_threads_k) {
            u64 from, to;
            span_of(s, side, i, from, to);
            if (to < from) values[1] |= SZS_PLAN_STATUS_DESCENDING;
            u64 const wide_length = to < from ? 0 : to - from;
            if (wide_length > 0xFFFFFFFFull) values[1] |= SZS_PLAN_STATUS_OVERFLOW;
            u32 const length = (u32)wide_length;
            values[0] = length > values[0] ? length : values[0];
            values[2] += length;
            values[3] += length ? (length + SZS_SYSTOLIC_BAND_ROWS - 1) / SZS_SYSTOLIC_BAND_ROWS : 1;
            values[4] += length ? (length + SZS_MYERS_CHAIN_BAND_ROWS - 1) / SZS_MYERS_CHAIN_BAND_ROWS : 1;
            u32 const slot = variant_slot(length, myers_words);
#pragma unroll
            for (u32 v = 0; v < SZS_PLAN_VARIANTS; ++v) values[5 + v] += slot == v;
        }
#pragma unroll
        for (int k = 0; k < plan_values_k; ++k) {
            u64 value = values[k];
#pragma unroll
            for (int offset = 32; offset >= 1; offset >>= 1) {
                u64 const other = shuffle_xor_u64(value, offset);
                value = k == 0 ? (other > value ? other : value) : k == 1 ? (value | other) : value + other;
            }
            if (lane == 0) wave_values[s][wave][k] = value;
        }
    }
    __syncthreads();
    if (tid < (u32)(sides * plan_values_k)) {
        int const s = tid / plan_values_k, k = tid % plan_values_k;
        u64 value = 0;
        for (int w = 0; w < plan_threads_k / 64; ++w) {
            u64 const other = wave_values[s][w][k];
            value = k == 0 ? (other > value ? other : value) : k == 1 ? (value | other) : value + other;
        }
        side_values[s][k] = value;
    }
    if (tid == 0) shared_cells = 0, shared_held = 0;
    __syncthreads();
    u32 const shared_status = (u32)(side_values[0][1] | (symmetric ? 0 : side_values[1][1]));
    u32 const shared_longest[2] = {(u32)side_values[0][0], (u32)side_values[symmetric ? 0 : 1][0]};

    // ---- symmetric calls: cells of the lower triangle = sum_i len_i * sum_{j <= i} len_j, in the caller's order
    if (symmetric) {
        u32 const chunk = (queries.count + plan_threads_k - 1) / plan_threads_k;
        u32 const first = tid * chunk < queries.count ? tid * chunk : queries.count;
        u32 const last = first + chunk < queries.count ? first + chunk : queries.count;
        u64 mine = 0;
        for (u32 i = first; i < last; ++i)
            mine += tape_offset(queries.offsets, queries.wide, (u64)i + 1) - tape_offset(queries.offsets, queries.wide, i);
        // prefix of the chunk sums: 64-bit, so two 32-bit scans would not do - a serial pass by one thread is 1024 adds
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
#pragma unroll
        for (int offset = 32; offset >= 1; offset >>= 1) cells += shuffle_xor_u64(cells, offset);
        __syncthreads();
        if (lane == 0) chunk_sums[wave] = cells;
        __syncthreads();
        if (tid == 0) {
            unsigned long long total = 0;
            for (int w = 0; w < plan_threads_k / 64; ++w) total += chunk_sums[w];
            shared_cells = total;
        }
        __syncthreads();
    }

    // ---- does this batch have the shape the host already enqueued launches for?
    if (tid == 0) {
        u32 held = expected.enabled && !shared_status;
        if (held) {
            int const query_side = symmetric ? 0 : (int)expected.query_side;
            for (u32 v = 0; v < SZS_PLAN_VARIANTS; ++v) held &= (u32)side_values[query_side][5 + v] == expected.variant_counts[v];
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
        for (u32 i = tid; i < side.count; i += plan_threads_k) {
            u64 from, to;
            span_of(s, side, i, from, to);
            atomicAdd(&histogram[(u32)(to - from)], 1u);
        }
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
            u64 from, to;
            span_of(s, side, i, from, to);
            u32 const length = (u32)(to - from);
            u32 const position = atomicAdd(&histogram[length], 1u); // equal lengths: any order scores the same matrix
            szs_string_ref_t ref;
            ref.address = side.base + from, ref.length = blank ? 0u : length, ref.index = i;
            side.ascending[position] = ref;
            side.descending[side.count - 1 - position] = ref;
        }
    }
    __syncthreads();

    if (tid == 0) { // one struct, written once: the host reads it after the stream has drained
        szs_plan_summary_t report;
        report.status = shared_status | (unsorted ? SZS_PLAN_STATUS_UNSORTED : 0u);
        report.speculation_held = shared_held;
        for (int s = 0; s < 2; ++s) {
            int const from = symmetric ? 0 : s;
            report.side[s].count = from ? candidates.count : queries.count;
            report.side[s].longest = (u32)side_values[from][0];
            report.side[s].symbols = side_values[from][2];
            report.side[s].bands_systolic = side_values[from][3];
            report.side[s].bands_chain = side_values[from][4];
            for (u32 v = 0; v < SZS_PLAN_VARIANTS; ++v) report.variant_counts[s][v] = (u32)side_values[from][5 + v];
        }
        report.symmetric_cells = shared_cells;
        report.sequence = expected.sequence; // the host can tell a fresh summary from a stale one
        *summary = report;
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
