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
 *    histogram   both sides at once: every string's length goes into its side's LDS histogram; offsets that descend,
 *                strings of 4 GiB and strings beyond the histogram are flagged, not scored (the host planner takes over);
 *    statistics  everything the host decides from - strings per bit-parallel launch variant, longest string, sums, the
 *                band counts of the tier model - is read off the histogram: 6 bins per thread, one scan, four reductions;
 *    check       the host may have enqueued the scoring launches ALREADY, shaped like the previous call of this engine
 *                (same launch variants, workspaces sized for the previous longest strings).  If this batch does not fit
 *                that shape, every ref is written with length 0 - the speculated launches then score empty strings,
 *                memory-safe and over in microseconds - and `speculation_held` stays 0; the host re-plans from the summary;
 *    scatter     counting sort: bins become positions, every string lands in TWO ref arrays, ascending (the lane side of
 *                every kernel) and descending (the workgroup side: longest first makes every launch variant a contiguous
 *                slice and hands out the heaviest workgroups first).
 *
 *  The summary lands in pinned host memory; the host reads it after the call's ONE synchronisation.
 *  Strings of `plan_bins_k` bytes or more are not sorted here (`status` says so; the host planner takes over).
 */
#include "device_common.hpp"

namespace szs_hip {

constexpr int plan_threads_k = 1024;
constexpr int plan_waves_k = plan_threads_k / 64;
constexpr u32 plan_staged_k = 4096; // refs of ONE side sorted through LDS before they are written out (64 KB)
constexpr u32 plan_cached_k = 4; // strings per thread and side kept in registers from the first fetch on
constexpr u32 plan_bins_k = SZS_PLAN_DEVICE_LONGEST + 1;         // lengths below this are counting-sorted in LDS
constexpr u32 plan_chunk_k = plan_bins_k / plan_threads_k;       // histogram bins per thread
static_assert(plan_bins_k % plan_threads_k == 0, "the histogram is cut into equal chunks");

__device__ __forceinline__ u64 tape_offset(void const *offsets, u32 wide, u64 index) {
    return wide ? static_cast<u64 const *>(offsets)[index] : (u64) static_cast<u32 const *>(offsets)[index];
}

/** Longest string (symbols) of launch variant `slot` 1 ... 9 (szs_hip_levenshtein_myers_round_words(): 8, 10 ... 64 words). */
__device__ __forceinline__ u32 variant_longest(u32 slot) {
    constexpr u32 words[SZS_PLAN_VARIANTS] = {0, SZS_MYERS_SHORT_WORDS, 10, 12, 16, 20, 24, 32, 48, 64};
    return 32u * words[slot];
}

__device__ __forceinline__ u64 shuffle_xor_u64(u64 value, int offset) {
    u32 const low = (u32)__shfl_xor((int)(u32)value, offset, 64), high = (u32)__shfl_xor((int)(u32)(value >> 32), offset, 64);
    return ((u64)high << 32) | low;
}

/**
 *  What bounds this kernel is LATENCY - dependent memory round trips, workgroup barriers, LDS crossbar shuffles - not work:
 *  a side of 1024 strings is one string per thread.  So both sides advance through every phase TOGETHER (two histograms,
 *  half the barriers), each thread keeps its strings' offsets in registers from the first load on, and every statistic the
 *  host wants - counts per launch variant, longest string, sums, band counts - is read off the HISTOGRAM of the lengths
 *  (6 bins per thread, then one scan and four reductions per side) instead of being reduced string by string (round 2's
 *  first version reduced fifteen 64-bit values per side through LDS atomics, which hipcc expands into scalar loops over
 *  the 64 lanes: 40 us; then through shuffles: 25 us).
 */
__global__ __launch_bounds__(plan_threads_k) void plan_kernel(szs_plan_side_t queries, szs_plan_side_t candidates,
                                                              int symmetric, u32 myers_words,
                                                              szs_plan_expectation_t expected,
                                                              szs_plan_summary_t *__restrict__ summary) {
    extern __shared__ __attribute__((aligned(16))) szs_string_ref_t staged[]; // `plan_staged_k` refs: phase 4
    __shared__ u32 histogram[2][plan_bins_k];
    __shared__ u32 wave_counts[2][plan_waves_k], wave_symbols[2][plan_waves_k], wave_bands_systolic[2][plan_waves_k],
        wave_bands_chain[2][plan_waves_k], wave_longest[2][plan_waves_k], wave_status[plan_waves_k];
    __shared__ u32 side_count_before_wave[2][plan_waves_k], side_symbols[2], side_bands_systolic[2], side_bands_chain[2],
        side_longest[2], side_strings[2], variants[2][SZS_PLAN_VARIANTS], shared_status, shared_held,
        rank_lengths[2][SZS_PLAN_RANK_SAMPLES + 1];
    __shared__ unsigned long long chunk_sums[plan_threads_k], shared_cells;

    u32 const tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    int const sides = symmetric ? 1 : 2;
    szs_plan_side_t const *const side_of[2] = {&queries, &candidates};
#ifdef SZS_PLAN_TIMESTAMPS // measuring aid (build variant): 100 MHz timestamps of the phases, behind the summary, for the trace
    unsigned long long *const stamps = reinterpret_cast<unsigned long long *>(summary) + 56;
#define SZS_PLAN_STAMP(K) do { if (tid == 0) stamps[K] = wall_clock64(); } while (0)
#else
#define SZS_PLAN_STAMP(K) do {} while (0)
#endif
    SZS_PLAN_STAMP(0);

    // ---- phase 0: both histograms cleared; each thread's first string of each side fetched (one round trip for the kernel
    //      when a side has at most 1024 strings: every later phase reads registers)
    // (round 4: the first FOUR strings of each side - a side of up to 4096 strings, config 5's 3163 - are fetched here, all
    // loads in flight at once: phases 1 and 4 walked a thread's strings one memory round trip after the other, 31 us for an
    // eighth of config 5u's 3,559 strings where 1,024 + 1,024 take 13)
    u64 first_from[2][plan_cached_k] = {}, first_to[2][plan_cached_k] = {}, first_start[2][plan_cached_k] = {};
    u32 first_length[2][plan_cached_k] = {};
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (u32 slot = 0; slot < plan_cached_k; ++slot) {
            u32 const i = tid + slot * plan_threads_k;
            if (s < sides && i < side_of[s]->count) {
                first_from[s][slot] = tape_offset(side_of[s]->offsets, side_of[s]->wide, i),
                first_to[s][slot] = tape_offset(side_of[s]->offsets, side_of[s]->wide, (u64)i + 1);
                if (side_of[s]->lengths) first_length[s][slot] = side_of[s]->lengths[i], first_start[s][slot] = side_of[s]->starts[i];
            }
        }
    for (int s = 0; s < sides; ++s)
#pragma unroll
        for (u32 k = 0; k < plan_chunk_k; ++k) histogram[s][k * plan_threads_k + tid] = 0;
    if (tid == 0) shared_cells = 0, shared_held = 0;
    __syncthreads();
    SZS_PLAN_STAMP(1);
    // a string of this thread: (from, to) of its span in the tape, its length in symbols and where its symbols live - from the
    // registers above for the first `plan_cached_k`, from memory beyond.  Codepoint engines: a string's length is its RUNE count
    // and its symbols live in the UTF-32 scratch tape (kernels.h).
    auto beyond = [&](int s, u32 i, u64 &from, u64 &to, u64 &length, u64 &address) {
        from = tape_offset(side_of[s]->offsets, side_of[s]->wide, i), to = tape_offset(side_of[s]->offsets, side_of[s]->wide, (u64)i + 1);
        length = !side_of[s]->lengths ? to - from : (u64)side_of[s]->lengths[i];
        address = !side_of[s]->lengths ? side_of[s]->base + from : side_of[s]->base + 4 * side_of[s]->starts[i];
    };
    auto cached = [&](int s, u32 slot, u64 &from, u64 &to, u64 &length, u64 &address) {
        from = first_from[s][slot], to = first_to[s][slot];
        length = !side_of[s]->lengths ? to - from : (u64)first_length[s][slot];
        address = !side_of[s]->lengths ? side_of[s]->base + from : side_of[s]->base + 4 * first_start[s][slot];
    };
    // every string of this thread, the cached ones first: `visit(s, i, from, to, length, address)`
    auto each_string = [&](int s, auto &&visit) {
#pragma unroll
        for (u32 slot = 0; slot < plan_cached_k; ++slot) {
            u32 const i = tid + slot * plan_threads_k;
            if (i >= side_of[s]->count) break;
            u64 from, to, length, address;
            cached(s, slot, from, to, length, address);
            visit(i, from, to, length, address);
        }
        for (u32 i = tid + plan_cached_k * plan_threads_k; i < side_of[s]->count; i += plan_threads_k) {
            u64 from, to, length, address;
            beyond(s, i, from, to, length, address);
            visit(i, from, to, length, address);
        }
    };

    // ---- phase 1: the histogram of the lengths; malformed or over-long strings only raise a flag (the host takes over)
    u32 status = 0;
#pragma unroll
    for (int s = 0; s < 2; ++s) // (unrolled: the cached strings are registers only under constant indices)
        if (s < sides) each_string(s, [&](u32, u64 from, u64 to, u64 length, u64) {
            if (to < from) status |= SZS_PLAN_STATUS_DESCENDING;
            else if (to - from > 0xFFFFFFFFull) status |= SZS_PLAN_STATUS_OVERFLOW;
            else if (length >= plan_bins_k) status |= SZS_PLAN_STATUS_UNSORTED;
            else atomicAdd(&histogram[s][(u32)length], 1u); // per-lane addresses: a plain ds_add
        });
#pragma unroll
    for (int offset = 32; offset >= 1; offset >>= 1) status |= (u32)__shfl_xor((int)status, offset, 64);
    if (lane == 0) wave_status[wave] = status;
    __syncthreads();
    SZS_PLAN_STAMP(2);

    // ---- phase 2: every statistic from the histogram.  Thread t owns bins [6 t, 6 t + 6) of both sides.
    u32 chunk_counts[2] = {0, 0}, chunk_inclusive[2] = {0, 0};
    for (int s = 0; s < sides; ++s) {
        u32 symbols = 0, bands_systolic = 0, bands_chain = 0, longest = 0;
#pragma unroll
        for (u32 k = 0; k < plan_chunk_k; ++k) {
            u32 const length = tid * plan_chunk_k + k, strings = histogram[s][length];
            chunk_counts[s] += strings;
            symbols += strings * length;
            bands_systolic += strings * (length ? (length + SZS_SYSTOLIC_BAND_ROWS - 1) / SZS_SYSTOLIC_BAND_ROWS : 1);
            bands_chain += strings * (length ? (length + SZS_MYERS_CHAIN_BAND_ROWS - 1) / SZS_MYERS_CHAIN_BAND_ROWS : 1);
            longest = strings ? length : longest;
        }
        u32 inclusive = chunk_counts[s];
#pragma unroll
        for (int offset = 1; offset < 64; offset <<= 1) {
            u32 const other = (u32)__shfl_up((int)inclusive, offset, 64);
            if (lane >= (u32)offset) inclusive += other;
        }
        chunk_inclusive[s] = inclusive;
#pragma unroll
        for (int offset = 32; offset >= 1; offset >>= 1) {
            symbols += (u32)__shfl_xor((int)symbols, offset, 64);
            bands_systolic += (u32)__shfl_xor((int)bands_systolic, offset, 64);
            bands_chain += (u32)__shfl_xor((int)bands_chain, offset, 64);
            u32 const other = (u32)__shfl_xor((int)longest, offset, 64);
            longest = other > longest ? other : longest;
        }
        if (lane == 63) wave_counts[s][wave] = inclusive;
        if (lane == 0)
            wave_symbols[s][wave] = symbols, wave_bands_systolic[s][wave] = bands_systolic, wave_bands_chain[s][wave] = bands_chain,
            wave_longest[s][wave] = longest;
    }
    __syncthreads();
    if (tid < (u32)sides) { // one thread per side folds the sixteen wavefronts
        int const s = (int)tid;
        u32 running = 0, symbols = 0, bands_systolic = 0, bands_chain = 0, longest = 0;
        for (int w = 0; w < plan_waves_k; ++w) {
            side_count_before_wave[s][w] = running, running += wave_counts[s][w];
            symbols += wave_symbols[s][w], bands_systolic += wave_bands_systolic[s][w], bands_chain += wave_bands_chain[s][w];
            longest = wave_longest[s][w] > longest ? wave_longest[s][w] : longest;
        }
        side_strings[s] = running, side_symbols[s] = symbols, side_bands_systolic[s] = bands_systolic, side_bands_chain[s] = bands_chain;
        side_longest[s] = longest;
    }
    if (tid == 2) {
        u32 all = 0;
        for (int w = 0; w < plan_waves_k; ++w) all |= wave_status[w];
        shared_status = all;
    }
    __syncthreads();

    SZS_PLAN_STAMP(3);
    // ---- phase 3: bins become positions (exclusive prefix); strings per launch variant are differences of positions
    for (int s = 0; s < sides; ++s) {
        u32 running = side_count_before_wave[s][wave] + chunk_inclusive[s] - chunk_counts[s];
#pragma unroll
        for (u32 k = 0; k < plan_chunk_k; ++k) {
            u32 const length = tid * plan_chunk_k + k, strings = histogram[s][length];
            histogram[s][length] = running, running += strings;
        }
    }
    __syncthreads();
    if (tid < (u32)(sides * SZS_PLAN_VARIANTS)) {
        int const s = tid / SZS_PLAN_VARIANTS;
        u32 const slot = tid % SZS_PLAN_VARIANTS, total = side_strings[s];
        auto strings_shorter_than = [&](u32 length) -> u32 { return length < plan_bins_k ? histogram[s][length] : total; };
        u32 strings;
        if (!myers_words) strings = slot == 0 ? total : 0; // weighted engines: one launch group
        else if (slot == 0) strings = total - strings_shorter_than(variant_longest(SZS_PLAN_VARIANTS - 1) + 1);
        else strings = strings_shorter_than(variant_longest(slot) + 1) - (slot == 1 ? 0 : strings_shorter_than(variant_longest(slot - 1) + 1));
        variants[s][slot] = strings;
    }
    // ---- the length at 33 ranks of each side (the queue order of hip/myers_queue.hip is planned from them, host/plan.c): strings
    //      of length l hold the ascending ranks [positions[l], positions[l + 1]), so the string at rank r is as long as the LARGEST
    //      l whose position is <= r - a binary search of the positions by one thread per sample, beside the threads above
    if (tid >= 64 && tid < 64 + (u32)sides * (SZS_PLAN_RANK_SAMPLES + 1)) {
        int const s = (int)((tid - 64) / (SZS_PLAN_RANK_SAMPLES + 1));
        u32 const k = (tid - 64) % (SZS_PLAN_RANK_SAMPLES + 1), total = side_strings[s];
        u32 length = 0;
        if (total) {
            u32 const rank = (u32)((u64)k * (total - 1) / SZS_PLAN_RANK_SAMPLES);
            u32 low = 0, high = plan_bins_k - 1;
            while (low < high) {
                u32 const middle = (low + high + 1) / 2;
                if (histogram[s][middle] <= rank) low = middle;
                else high = middle - 1;
            }
            length = low;
        }
        rank_lengths[s][k] = length;
    }
    __syncthreads();

    // ---- symmetric calls: cells of the lower triangle = sum_i len_i * sum_{j <= i} len_j, in the caller's order
    if (symmetric && !shared_status) {
        auto symbols_of_query = [&](u32 i) -> u64 { // (consecutive strings per thread here: straight from memory)
            return queries.lengths ? (u64)queries.lengths[i] : tape_offset(queries.offsets, queries.wide, (u64)i + 1) - tape_offset(queries.offsets, queries.wide, i);
        };
        u32 const chunk = (queries.count + plan_threads_k - 1) / plan_threads_k;
        u32 const first = tid * chunk < queries.count ? tid * chunk : queries.count;
        u32 const last = first + chunk < queries.count ? first + chunk : queries.count;
        u64 mine = 0;
        for (u32 i = first; i < last; ++i)
            mine += symbols_of_query(i);
        chunk_sums[tid] = mine; // prefix of the chunk sums: 64-bit; a serial pass by one thread is 1024 additions
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
            u64 const length = symbols_of_query(i);
            running += length, cells += length * running;
        }
#pragma unroll
        for (int offset = 32; offset >= 1; offset >>= 1) cells += shuffle_xor_u64(cells, offset);
        __syncthreads();
        if (lane == 0) chunk_sums[wave] = cells;
        __syncthreads();
        if (tid == 0) {
            unsigned long long total = 0;
            for (int w = 0; w < plan_waves_k; ++w) total += chunk_sums[w];
            shared_cells = total;
        }
        __syncthreads();
    }

    SZS_PLAN_STAMP(4);
    // ---- does this batch have the shape the host already enqueued launches for?
    if (tid == 0) {
        u32 held = expected.enabled && !shared_status;
        if (held) {
            int const query_side = symmetric ? 0 : (int)expected.query_side;
            for (u32 v = 0; v < SZS_PLAN_VARIANTS; ++v) held &= variants[query_side][v] == expected.variant_counts[v];
            held &= side_longest[0] <= expected.longest[0];
            held &= side_longest[symmetric ? 0 : 1] <= expected.longest[1];
        }
        // codepoints: every string was transcoded (none skipped for want of room) and the arrays hold what the launches index with
        if (held && expected.runes_needed) held &= *expected.runes_needed <= expected.runes_capacity;
        if (held && expected.alphabet)
            held &= expected.alphabet_flags[0] != 0 && expected.alphabet_flags[2] == 0 && expected.alphabet_flags[1] <= expected.alphabet;
        shared_held = held;
    }
    __syncthreads();
    bool const blank = expected.enabled && !shared_held; // speculated launches must find nothing to score
    SZS_PLAN_STAMP(5);

    // ---- phase 4: scatter into the ascending and the descending ref arrays
    if (shared_status) { // nothing was sorted; launches that are already in flight must still find only empty strings
        if (expected.enabled)
            for (int s = 0; s < sides; ++s)
                for (u32 i = tid; i < side_of[s]->count; i += plan_threads_k) {
                    szs_string_ref_t ref;
                    ref.address = side_of[s]->base, ref.length = 0, ref.index = i;
                    side_of[s]->ascending[i] = ref, side_of[s]->descending[i] = ref;
                }
    }
    else
#pragma unroll
        for (int s = 0; s < 2; ++s)
            if (s < sides) {
                // (round 4) A ref lands at the position its length sorts it to - sixteen bytes at a random place, twice: 64
                // write transactions per store instruction, all through ONE compute unit (17.9 of the planner's 44 us on config
                // 5's 6,326 strings, 4 of 13 on config 2's 2,048).  A side of up to `plan_staged_k` strings is sorted into LDS
                // instead and leaves it in order: position p and count - 1 - p, whole lines per wavefront.
                u32 const count = side_of[s]->count;
                bool const through_lds = count <= plan_staged_k;
                each_string(s, [&](u32 i, u64, u64, u64 symbols, u64 address) {
                    u32 const length = (u32)symbols;
                    u32 const position = atomicAdd(&histogram[s][length], 1u); // equal lengths: any order scores the same matrix
                    szs_string_ref_t ref;
                    ref.address = address, ref.length = blank ? 0u : length, ref.index = i;
                    if (through_lds) staged[position] = ref;
                    else side_of[s]->ascending[position] = ref, side_of[s]->descending[count - 1 - position] = ref;
                });
                if (through_lds) {
                    __syncthreads();
                    for (u32 position = tid; position < count; position += plan_threads_k) {
                        szs_string_ref_t const ref = staged[position];
                        side_of[s]->ascending[position] = ref, side_of[s]->descending[count - 1 - position] = ref;
                    }
                    __syncthreads(); // the other side sorts into the same LDS
                }
            }

    SZS_PLAN_STAMP(6);
    if (tid == 0) { // one struct, written once: the host reads it after the stream has drained
        szs_plan_summary_t report;
        report.status = shared_status;
        report.speculation_held = shared_held;
        for (int s = 0; s < 2; ++s) {
            int const from = symmetric ? 0 : s;
            report.side[s].count = from ? candidates.count : queries.count;
            report.side[s].longest = side_longest[from];
            report.side[s].symbols = side_symbols[from];
            report.side[s].bands_systolic = side_bands_systolic[from];
            report.side[s].bands_chain = side_bands_chain[from];
            for (u32 v = 0; v < SZS_PLAN_VARIANTS; ++v) report.variant_counts[s][v] = variants[from][v];
            for (u32 k = 0; k <= SZS_PLAN_RANK_SAMPLES; ++k) report.rank_lengths[s][k] = rank_lengths[from][k];
        }
        report.symmetric_cells = shared_cells;
        report.sequence = expected.sequence; // the host can tell a fresh summary from a stale one
        *summary = report;
    }
    SZS_PLAN_STAMP(7);
#undef SZS_PLAN_STAMP
}

} // namespace szs_hip

extern "C" int szs_hip_plan(szs_plan_side_t const *queries, szs_plan_side_t const *candidates, unsigned myers_words,
                            szs_plan_expectation_t const *expected, szs_plan_summary_t *summary, void *stream) {
    using namespace szs_hip;
    szs_plan_expectation_t none = {};
    // 64 KB of dynamic LDS beside the 58 KB of histograms: asked for once per device
    static int allowed_on[device_slots_k];
    int *const slot = &allowed_on[device_slot()];
    if (!cached(slot)) {
        if (hipFuncSetAttribute(reinterpret_cast<void const *>(plan_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)(plan_staged_k * sizeof(szs_string_ref_t))) != hipSuccess)
            return (int)hipGetLastError();
        remember(slot, 1);
    }
    hipLaunchKernelGGL(plan_kernel, dim3(1), dim3(plan_threads_k), plan_staged_k * sizeof(szs_string_ref_t), static_cast<hipStream_t>(stream), *queries,
                       candidates ? *candidates : *queries, candidates ? 0 : 1, (u32)myers_words, expected ? *expected : none,
                       summary);
    return (int)hipGetLastError();
}
