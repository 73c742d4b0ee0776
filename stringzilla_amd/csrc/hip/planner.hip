/*
 *  planner.hip - the O(Q + C) planner on the DEVICE: tape offsets in, length-sorted string refs out, no host round trip.
 *
 *  The reference plans per CELL on the device - a 112-byte task per (query, candidate) pair, counting-sorted by size tier
 *  and scattered afterwards (/root/reference/include/stringzillas/similarities/cuda.cuh:1652-1711,1887-1957) - or, on its
 *  fast path, skips task materialisation altogether (cuda.cuh:4297-4340).  This build plans per ROW and COLUMN
 *  (host/plan.c); round 1 did that on the host, which cost a download of the offsets and a stream synchronisation before
 *  the first scoring launch could even be enqueued (18 % of config 2's wall time).  Here the same plan is produced by one
 *  workgroup of 1024 threads PER SIDE straight from the caller's offsets:
 *
 *    histogram   every string's length goes into its side's LDS histogram; offsets that descend,
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
 *  a side of 1024 strings is one string per thread.  Each thread keeps its strings' offsets in registers from the first load
 *  on, and every statistic the host wants - counts per launch variant, longest string, sums, band counts - is read off the
 *  HISTOGRAM of the lengths (6 bins per thread, then one scan and four reductions) instead of being reduced string by string.
 *
 *  Round 5: ONE WORKGROUP PER SIDE.  Rounds 2-4 planned both sides in one workgroup, phase by phase together; the two sides
 *  share nothing but the verdict on the host's expectation, so each now has a workgroup (and a CU) of its own - half the
 *  strings per thread, half the LDS traffic per phase - and the SECOND one to finish (a counter in device memory says which)
 *  folds the two verdicts into the summary: config-5-sized sides (2 x 3163 strings) 33-37 us -> see profiles/r05.  A side that
 *  does not fit the expectation blanks ITS OWN refs; a speculated launch then scores against empty strings whichever side it was.
 */
__global__ __launch_bounds__(plan_threads_k) void plan_kernel(szs_plan_side_t queries, szs_plan_side_t candidates,
                                                              int symmetric, u32 myers_words,
                                                              szs_plan_expectation_t expected,
                                                              szs_plan_summary_t *__restrict__ summary, u32 *__restrict__ verdicts) {
    extern __shared__ __attribute__((aligned(16))) szs_string_ref_t staged[]; // `plan_staged_k` refs: phase 4
    __shared__ u32 histogram[plan_bins_k];
    __shared__ u32 wave_counts[plan_waves_k], wave_symbols[plan_waves_k], wave_bands_systolic[plan_waves_k],
        wave_bands_chain[plan_waves_k], wave_longest[plan_waves_k], wave_status[plan_waves_k];
    __shared__ u32 count_before_wave[plan_waves_k], side_symbols, side_bands_systolic, side_bands_chain, side_longest, side_strings,
        variants[SZS_PLAN_VARIANTS], shared_status, shared_held, rank_lengths[SZS_PLAN_RANK_SAMPLES + 1];
    __shared__ unsigned long long chunk_sums[plan_threads_k], shared_cells;

    u32 const tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    int const mine = symmetric ? 0 : (int)blockIdx.x; // the side this workgroup plans: 0 the caller's queries, 1 its candidates
    szs_plan_side_t const &side = mine ? candidates : queries;
#ifdef SZS_PLAN_TIMESTAMPS // measuring aid (build variant): 100 MHz timestamps of the phases, behind the summary, for the trace
    unsigned long long *const stamps = reinterpret_cast<unsigned long long *>(summary) + 56;
#define SZS_PLAN_STAMP(K) do { if (tid == 0 && mine == 0) stamps[K] = wall_clock64(); } while (0)
#else
#define SZS_PLAN_STAMP(K) do {} while (0)
#endif
    SZS_PLAN_STAMP(0);

    // ---- phase 0: the histogram cleared; each thread's first FOUR strings fetched, all loads in flight at once (a side of up to
    //      4096 strings - config 5's 3163 - is then one round trip; every later phase reads registers)
    u64 first_from[plan_cached_k] = {}, first_to[plan_cached_k] = {}, first_start[plan_cached_k] = {};
    u32 first_length[plan_cached_k] = {};
#pragma unroll
    for (u32 slot = 0; slot < plan_cached_k; ++slot) {
        u32 const i = tid + slot * plan_threads_k;
        if (i < side.count) {
            first_from[slot] = tape_offset(side.offsets, side.wide, i), first_to[slot] = tape_offset(side.offsets, side.wide, (u64)i + 1);
            if (side.lengths) first_length[slot] = side.lengths[i], first_start[slot] = side.starts[i];
        }
    }
#pragma unroll
    for (u32 k = 0; k < plan_chunk_k; ++k) histogram[k * plan_threads_k + tid] = 0;
    if (tid == 0) shared_cells = 0, shared_held = 0;
    __syncthreads();
    SZS_PLAN_STAMP(1);
    // a string of this thread: (from, to) of its span in the tape, its length in symbols and where its symbols live - from the
    // registers above for the first `plan_cached_k`, from memory beyond.  Codepoint engines: a string's length is its RUNE count
    // and its symbols live in the UTF-32 scratch tape (kernels.h).
    auto beyond = [&](u32 i, u64 &from, u64 &to, u64 &length, u64 &address) {
        from = tape_offset(side.offsets, side.wide, i), to = tape_offset(side.offsets, side.wide, (u64)i + 1);
        length = !side.lengths ? to - from : (u64)side.lengths[i];
        address = !side.lengths ? side.base + from : side.base + 4 * side.starts[i];
    };
    auto cached = [&](u32 slot, u64 &from, u64 &to, u64 &length, u64 &address) {
        from = first_from[slot], to = first_to[slot];
        length = !side.lengths ? to - from : (u64)first_length[slot];
        address = !side.lengths ? side.base + from : side.base + 4 * first_start[slot];
    };
    // every string of this thread, the cached ones first: `visit(i, from, to, length, address)`
    auto each_string = [&](auto &&visit) {
#pragma unroll
        for (u32 slot = 0; slot < plan_cached_k; ++slot) {
            u32 const i = tid + slot * plan_threads_k;
            if (i >= side.count) break;
            u64 from, to, length, address;
            cached(slot, from, to, length, address);
            visit(i, from, to, length, address);
        }
        for (u32 i = tid + plan_cached_k * plan_threads_k; i < side.count; i += plan_threads_k) {
            u64 from, to, length, address;
            beyond(i, from, to, length, address);
            visit(i, from, to, length, address);
        }
    };

    // ---- phase 1: the histogram of the lengths; malformed or over-long strings only raise a flag (the host takes over)
    u32 status = 0;
    each_string([&](u32, u64 from, u64 to, u64 length, u64) {
        if (to < from) status |= SZS_PLAN_STATUS_DESCENDING;
        else if (to - from > 0xFFFFFFFFull) status |= SZS_PLAN_STATUS_OVERFLOW;
        else if (length >= plan_bins_k) status |= SZS_PLAN_STATUS_UNSORTED;
        else atomicAdd(&histogram[(u32)length], 1u); // per-lane addresses: a plain ds_add
    });
#pragma unroll
    for (int offset = 32; offset >= 1; offset >>= 1) status |= (u32)__shfl_xor((int)status, offset, 64);
    if (lane == 0) wave_status[wave] = status;
    __syncthreads();
    SZS_PLAN_STAMP(2);

    // ---- phase 2: every statistic from the histogram.  Thread t owns bins [6 t, 6 t + 6).
    u32 chunk_counts = 0, chunk_inclusive = 0;
    {
        u32 symbols = 0, bands_systolic = 0, bands_chain = 0, longest = 0;
#pragma unroll
        for (u32 k = 0; k < plan_chunk_k; ++k) {
            u32 const length = tid * plan_chunk_k + k, strings = histogram[length];
            chunk_counts += strings;
            symbols += strings * length;
            bands_systolic += strings * (length ? (length + SZS_SYSTOLIC_BAND_ROWS - 1) / SZS_SYSTOLIC_BAND_ROWS : 1);
            bands_chain += strings * (length ? (length + SZS_MYERS_CHAIN_BAND_ROWS - 1) / SZS_MYERS_CHAIN_BAND_ROWS : 1);
            longest = strings ? length : longest;
        }
        u32 inclusive = chunk_counts;
#pragma unroll
        for (int offset = 1; offset < 64; offset <<= 1) {
            u32 const other = (u32)__shfl_up((int)inclusive, offset, 64);
            if (lane >= (u32)offset) inclusive += other;
        }
        chunk_inclusive = inclusive;
#pragma unroll
        for (int offset = 32; offset >= 1; offset >>= 1) {
            symbols += (u32)__shfl_xor((int)symbols, offset, 64);
            bands_systolic += (u32)__shfl_xor((int)bands_systolic, offset, 64);
            bands_chain += (u32)__shfl_xor((int)bands_chain, offset, 64);
            u32 const other = (u32)__shfl_xor((int)longest, offset, 64);
            longest = other > longest ? other : longest;
        }
        if (lane == 63) wave_counts[wave] = inclusive;
        if (lane == 0)
            wave_symbols[wave] = symbols, wave_bands_systolic[wave] = bands_systolic, wave_bands_chain[wave] = bands_chain,
            wave_longest[wave] = longest;
    }
    __syncthreads();
    if (tid == 0) { // one thread folds the sixteen wavefronts
        u32 running = 0, symbols = 0, bands_systolic = 0, bands_chain = 0, longest = 0;
        for (int w = 0; w < plan_waves_k; ++w) {
            count_before_wave[w] = running, running += wave_counts[w];
            symbols += wave_symbols[w], bands_systolic += wave_bands_systolic[w], bands_chain += wave_bands_chain[w];
            longest = wave_longest[w] > longest ? wave_longest[w] : longest;
        }
        side_strings = running, side_symbols = symbols, side_bands_systolic = bands_systolic, side_bands_chain = bands_chain;
        side_longest = longest;
    }
    if (tid == 64) {
        u32 all = 0;
        for (int w = 0; w < plan_waves_k; ++w) all |= wave_status[w];
        shared_status = all;
    }
    __syncthreads();

    SZS_PLAN_STAMP(3);
    // ---- phase 3: bins become positions (exclusive prefix); strings per launch variant are differences of positions
    {
        u32 running = count_before_wave[wave] + chunk_inclusive - chunk_counts;
#pragma unroll
        for (u32 k = 0; k < plan_chunk_k; ++k) {
            u32 const length = tid * plan_chunk_k + k, strings = histogram[length];
            histogram[length] = running, running += strings;
        }
    }
    __syncthreads();
    if (tid < SZS_PLAN_VARIANTS) {
        u32 const slot = tid, total = side_strings;
        auto strings_shorter_than = [&](u32 length) -> u32 { return length < plan_bins_k ? histogram[length] : total; };
        u32 strings;
        if (!myers_words) strings = slot == 0 ? total : 0; // weighted engines: one launch group
        else if (slot == 0) strings = total - strings_shorter_than(variant_longest(SZS_PLAN_VARIANTS - 1) + 1);
        else strings = strings_shorter_than(variant_longest(slot) + 1) - (slot == 1 ? 0 : strings_shorter_than(variant_longest(slot - 1) + 1));
        variants[slot] = strings;
    }
    // ---- the length at 33 ranks of the side (the queue order of hip/myers_queue.hip is planned from them, host/plan.c): strings
    //      of length l hold the ascending ranks [positions[l], positions[l + 1]), so the string at rank r is as long as the LARGEST
    //      l whose position is <= r - a binary search of the positions by one thread per sample, beside the threads above
    if (tid >= 64 && tid < 64 + (SZS_PLAN_RANK_SAMPLES + 1)) {
        u32 const k = tid - 64, total = side_strings;
        u32 length = 0;
        if (total) {
            u32 const rank = (u32)((u64)k * (total - 1) / SZS_PLAN_RANK_SAMPLES);
            u32 low = 0, high = plan_bins_k - 1;
            while (low < high) {
                u32 const middle = (low + high + 1) / 2;
                if (histogram[middle] <= rank) low = middle;
                else high = middle - 1;
            }
            length = low;
        }
        rank_lengths[k] = length;
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
        u64 own = 0;
        for (u32 i = first; i < last; ++i)
            own += symbols_of_query(i);
        chunk_sums[tid] = own; // prefix of the chunk sums: 64-bit; a serial pass by one thread is 1024 additions
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
    // ---- does this SIDE of the batch have the shape the host already enqueued launches for?  (the whole batch does when both do)
    if (tid == 0) {
        u32 held = expected.enabled && !shared_status;
        if (held) {
            int const query_side = symmetric ? 0 : (int)expected.query_side;
            if (mine == query_side)
                for (u32 v = 0; v < SZS_PLAN_VARIANTS; ++v) held &= variants[v] == expected.variant_counts[v];
            held &= side_longest <= expected.longest[mine];
            if (symmetric) held &= side_longest <= expected.longest[1];
        }
        // codepoints: every string was transcoded (none skipped for want of room) and the arrays hold what the launches index with
        // (facts about the whole batch: the queries' workgroup checks them)
        if (held && mine == 0 && expected.runes_needed) held &= *expected.runes_needed <= expected.runes_capacity;
        if (held && mine == 0 && expected.alphabet)
            held &= expected.alphabet_flags[0] != 0 && expected.alphabet_flags[2] == 0 && expected.alphabet_flags[1] <= expected.alphabet;
        shared_held = held;
    }
    __syncthreads();
    bool const blank = expected.enabled && !shared_held; // speculated launches must find nothing to score
    SZS_PLAN_STAMP(5);

    // ---- phase 4: scatter into the ascending and the descending ref arrays
    if (shared_status) { // nothing was sorted; launches that are already in flight must still find only empty strings
        if (expected.enabled)
            for (u32 i = tid; i < side.count; i += plan_threads_k) {
                szs_string_ref_t ref;
                ref.address = side.base, ref.length = 0, ref.index = i;
                side.ascending[i] = ref, side.descending[i] = ref;
            }
    }
    else {
        // A ref lands at the position its length sorts it to - sixteen bytes at a random place, twice: 64 write transactions
        // per store instruction, all through ONE compute unit.  A side of up to `plan_staged_k` strings is sorted into LDS
        // instead and leaves it in order: position p and count - 1 - p, whole lines per wavefront.
        u32 const count = side.count;
        bool const through_lds = count <= plan_staged_k;
        each_string([&](u32 i, u64, u64, u64 symbols, u64 address) {
            u32 const length = (u32)symbols;
            u32 const position = atomicAdd(&histogram[length], 1u); // equal lengths: any order scores the same matrix
            szs_string_ref_t ref;
            ref.address = address, ref.length = blank ? 0u : length, ref.index = i;
            if (through_lds) staged[position] = ref;
            else side.ascending[position] = ref, side.descending[count - 1 - position] = ref;
        });
        if (through_lds) {
            __syncthreads();
            for (u32 position = tid; position < count; position += plan_threads_k) {
                szs_string_ref_t const ref = staged[position];
                side.ascending[position] = ref, side.descending[count - 1 - position] = ref;
            }
        }
    }

    SZS_PLAN_STAMP(6);
    // ---- the summary (pinned host memory; the host reads it after the stream has drained): this side's part by this workgroup,
    //      the verdict on the whole batch by whichever workgroup finishes second
    if (tid <= SZS_PLAN_RANK_SAMPLES) {
        summary->rank_lengths[mine][tid] = rank_lengths[tid];
        if (symmetric) summary->rank_lengths[1][tid] = rank_lengths[tid];
    }
    if (tid < SZS_PLAN_VARIANTS) {
        summary->variant_counts[mine][tid] = variants[tid];
        if (symmetric) summary->variant_counts[1][tid] = variants[tid];
    }
    if (tid == 0) {
        szs_side_stats_t stats;
        stats.count = side.count, stats.longest = side_longest, stats.symbols = side_symbols;
        stats.bands_systolic = side_bands_systolic, stats.bands_chain = side_bands_chain;
        summary->side[mine] = stats;
        if (symmetric) summary->side[1] = stats;
        u32 all_status = shared_status, all_held = shared_held;
        bool last = true;
        if (!symmetric) { // two workgroups: leave this one's verdict in device memory; the second to arrive folds both
            verdicts[2 + 2 * mine] = shared_status, verdicts[3 + 2 * mine] = shared_held;
            __threadfence();
            last = (atomicAdd(&verdicts[0], 1u) & 1u) != 0; // (counted up for ever: the parity tells the second from the first)
            if (last) {
                __threadfence();
                u32 const other_status = __hip_atomic_load(&verdicts[2 + 2 * (1 - mine)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                u32 const other_held = __hip_atomic_load(&verdicts[3 + 2 * (1 - mine)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                all_status |= other_status, all_held &= other_held;
            }
        }
        if (last) {
            summary->status = all_status, summary->speculation_held = all_held;
            summary->symmetric_cells = shared_cells;
            __threadfence_system();
            summary->sequence = expected.sequence; // the host can tell a fresh summary from a stale one
        }
    }
    SZS_PLAN_STAMP(7);
#undef SZS_PLAN_STAMP
}

} // namespace szs_hip

extern "C" int szs_hip_plan(szs_plan_side_t const *queries, szs_plan_side_t const *candidates, unsigned myers_words,
                            szs_plan_expectation_t const *expected, szs_plan_summary_t *summary, uint32_t *verdicts, void *stream) {
    using namespace szs_hip;
    szs_plan_expectation_t none = {};
    if (candidates && !verdicts) return (int)hipErrorInvalidValue;
    // 64 KB of dynamic LDS beside the 29 KB of histogram: asked for once per device
    static int allowed_on[device_slots_k];
    int *const slot = &allowed_on[device_slot()];
    if (!cached(slot)) {
        if (hipFuncSetAttribute(reinterpret_cast<void const *>(plan_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)(plan_staged_k * sizeof(szs_string_ref_t))) != hipSuccess)
            return (int)hipGetLastError();
        remember(slot, 1);
    }
    hipLaunchKernelGGL(plan_kernel, dim3(candidates ? 2 : 1), dim3(plan_threads_k), plan_staged_k * sizeof(szs_string_ref_t), static_cast<hipStream_t>(stream),
                       *queries, candidates ? *candidates : *queries, candidates ? 0 : 1, (u32)myers_words, expected ? *expected : none, summary, verdicts);
    return (int)hipGetLastError();
}
