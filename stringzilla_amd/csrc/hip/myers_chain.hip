/*
 *  myers_chain.hip - unit-cost byte-level Levenshtein for FEW, LONG pairs on gfx950: the bit-parallel recurrence of
 *  lev_myers.hip laid out as the band chain of systolic.hip.
 *
 *  Fills, for unit costs, the slot of the reference's cooperative and tiled tiers
 *      unit_myers_multiword_cooperative_per_cuda_warp_   /root/reference/include/stringzillas/similarities/cuda.cuh:2767
 *      score_across_cuda_device_                          .../similarities/cuda.cuh:729   (what pairs beyond 2048 B fall to)
 *  and must return exactly what the serial scorer returns (levenshtein_distance_myers, serial.hpp:2073-2314) - here in
 *  its own BLOCK form (serial.hpp:2182-2204): the add is per word, only the horizontal +-1 deltas ripple between words.
 *
 *  lev_myers.hip keeps a whole pattern (<= 2048 bytes) in the registers of ONE lane and needs thousands of pairs to fill
 *  the device; systolic.hip spreads a pair over wavefronts but scores one cell per few instructions.  This kernel does
 *  both: 32 cells per ~19 instructions AND one pair per chain of wavefronts.
 *
 *  - A BAND is 64 lanes x 32 pattern rows = 2048 rows = one wavefront; lane l owns word l of the band's bit-vector:
 *    VP / VN of rows [32 l, 32 l + 32).  The match masks Peq[symbol][lane] of the band live in LDS (64 KB, conflict-free
 *    ds_read_b32: the lanes of a wavefront read consecutive dwords).  They depend on the QUERY only, so the four
 *    wavefronts of a workgroup score four different candidates against the same query band and share one table -
 *    which is what lets 8 wavefronts live on a CU instead of 2 when a batch is large.
 *  - A lane advances K = 8 text columns per step.  Between words only the horizontal deltas of a word's LAST row travel:
 *    2 bits per column, so ONE dword per step (`v_mov_b32_dpp wave_shr:1`) carries everything lane l + 1 needs from
 *    lane l; the K text bytes travel the same way one step ahead, so their masks are fetched from LDS a step early.
 *  - Bands of a pair are chained through memory exactly like systolic.hip's: lane 63 parks one dword per step, publishes
 *    an epoch-tagged progress word every 16 steps (agent-scope accesses, no cache maintenance), tickets are drawn in
 *    (pair, band) order when a wavefront starts, a stalled wait flags the call instead of hanging the device.
 *  - No per-column score: when a lane has consumed the text, rows contribute popcount(VP) - popcount(VN); a band sums
 *    its lanes, publishes the partial sum, and the band that finishes last adds them up: distance = len(text) + sum.
 */
#include "device_common.hpp"

namespace szs_hip {

constexpr u32 chain_band_rows_k = 64u * 32u;    // pattern rows per band = per wavefront
constexpr u32 chain_columns_k = 8;              // K: text columns a lane advances per step
constexpr u32 chain_chunk_steps_k = 16;         // steps per hand-over between bands (128 columns)
constexpr u32 chain_slack_words_k = 64;         // parked words past the longest candidate
constexpr size_t chain_header_bytes_k = 256;    // ticket counter [0] and stall flag [1]: the layout systolic.hip uses
constexpr unsigned long long chain_patience_ticks_k = 200000000ull; // 2 s of the 100 MHz wall clock: how long a band waits for its predecessor before it flags the call (the host then re-runs it on the lanes tier)
constexpr u32 chain_max_waves_k = 16;           // wavefronts per workgroup: up to 16 CANDIDATES against the same query band
static_assert(chain_band_rows_k == SZS_MYERS_CHAIN_BAND_ROWS, "the host planner models bands of this height");
static_assert(chain_columns_k == 8, "the text bytes of one step travel in two dwords, the deltas in one");

__device__ __forceinline__ u32 chain_from_lane_above(u32 first, u32 value) { // lane 0 receives `first`
    return (u32)__builtin_amdgcn_update_dpp((int)first, (int)value, 0x138 /* wave_shr:1 */, 0xF, 0xF, false);
}
__device__ __forceinline__ u32 chain_load(u32 const *cell) {
    return __hip_atomic_load(cell, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void chain_store(u32 *cell, u32 value) {
    __hip_atomic_store(cell, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ i32 wave_sum_i32(i32 value) {
#pragma unroll
    for (int offset = 32; offset >= 1; offset >>= 1) value += __shfl_xor(value, offset, 64);
    return value;
}

/**
 *  One 32-row word, one text column (Hyyro's block step; serial.hpp:2182-2204).
 *  `hp_in` / `hn_in`: the horizontal delta entering the word's first row (+1 / -1) as 0/1 values;
 *  returns the delta leaving its last row as (hp_out | hn_out << 1).
 */
__device__ __forceinline__ u32 chain_word_step(u32 &vp, u32 &vn, u32 eq, u32 hp_in, u32 hn_in) {
    u32 const xv = eq | vn;
    eq |= hn_in;
    u32 const xh = (((eq & vp) + vp) ^ vp) | eq;
    u32 hp = vn | ~(xh | vp);
    u32 hn = vp & xh;
    u32 const out = (hp >> 31) | ((hn >> 31) << 1);
    hp = (hp << 1) | hp_in;
    hn = (hn << 1) | hn_in;
    vp = hn | ~(xv | hp);
    vn = hp & xv;
    return out;
}

/**
 *  Control block (epoch-tagged 64-bit words, zeroed once at allocation - see hip/kernels.h):
 *    work_counter[0], [1]                ticket counter, stall flag
 *    progress[pair * max_bands + band]   text columns whose last-row deltas that band has parked
 *    partial[pair * max_bands + band]    that band's sum of popcount(VP) - popcount(VN) at the end of the text
 *    done[pair]                          bands of the pair that have published their partial sum
 *  parked[pair][step]: one dword per step, bit 2j / 2j+1 = the +1 / -1 delta under the band's last row at column K step + j.
 */
template <u32 chain_waves_k>
__global__ __launch_bounds__(64 * chain_waves_k) void myers_chain_kernel(szs_string_ref_t const *__restrict__ queries, u32 queries_count,
                                                        szs_string_ref_t const *__restrict__ candidates,
                                                        u32 candidates_count, u32 max_bands, u64 *__restrict__ results,
                                                        u64 results_row_stride, int layout_flags,
                                                        u64 *__restrict__ work_counter, u64 *progress, u64 *partial,
                                                        u64 *done, u32 *parked, u32 parked_words, u32 epoch) {
    constexpr u32 K = chain_columns_k, chunk_steps = chain_chunk_steps_k;
    __shared__ u32 peq[256 * 64]; // [symbol][lane]: the whole 64 KB a workgroup may declare statically

    u32 const lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    u32 const candidate_groups = (candidates_count + chain_waves_k - 1) / chain_waves_k;
    u64 const total_tickets = (u64)queries_count * candidate_groups * max_bands;
    u64 const tag = (u64)epoch << 32;

    // One ticket per WORKGROUP, in (query, candidate group, band) order; the table doubles as the mailbox that hands the
    // ticket to the other wavefronts before it is built.
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_max(work_counter, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        peq[0] = (u32)__hip_atomic_fetch_add(work_counter, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    u32 const ticket = __builtin_amdgcn_readfirstlane(peq[0]);
    __syncthreads();
    if (ticket >= total_tickets) return;
    u32 const band = ticket % max_bands, group = (ticket / max_bands) % candidate_groups;
    szs_string_ref_t const query = queries[ticket / max_bands / candidate_groups];
    u32 const m = query.length;
    u32 const bands = m ? (m + chain_band_rows_k - 1) / chain_band_rows_k : 1;
    if (band >= bands) return; // the whole workgroup: nothing below depends on the candidate yet

    // ---- Peq of this query band, built by all 256 threads: thread t owns the rows t, t + 256, ... of the band
    for (u32 i = threadIdx.x; i < 256 * 64; i += 64 * chain_waves_k) peq[i] = 0;
    __syncthreads();
    {
        u32 const band_first_row = band * chain_band_rows_k;
        u32 const band_rows = m - band_first_row < chain_band_rows_k ? m - band_first_row : chain_band_rows_k;
        u8 const *const pattern = reinterpret_cast<u8 const *>(query.address) + band_first_row;
        for (u32 r = threadIdx.x; r < band_rows; r += 64 * chain_waves_k) atomicOr(&peq[(u32)pattern[r] * 64 + r / 32], 1u << (r % 32));
    }
    __syncthreads(); // the only barriers: from here on every wavefront runs on its own

    // ---- this wavefront's candidate
    u32 const candidate_slot = group * chain_waves_k + wave;
    if (candidate_slot >= candidates_count) return;
    szs_string_ref_t const candidate = candidates[candidate_slot];
    if ((layout_flags & SZS_LAYOUT_SYMMETRIC) && candidate.index > query.index) return;
    u32 const n = candidate.length;
    u32 const pair = (ticket / max_bands / candidate_groups) * candidates_count + candidate_slot;

    auto write_result = [&](u64 distance) {
        bool const transposed = (layout_flags & SZS_LAYOUT_TRANSPOSED) != 0;
        u64 const row = transposed ? candidate.index : query.index, column = transposed ? query.index : candidate.index;
        results[row * results_row_stride + column] = distance;
        if ((layout_flags & SZS_LAYOUT_SYMMETRIC) && candidate.index != query.index)
            results[column * results_row_stride + row] = distance;
    };
    if (m == 0 || n == 0) { // the distance to an empty string is the other one's length
        if (lane == 0 && band == 0) write_result(m ? m : n);
        return;
    }

    bool const first_band = band == 0, last_band = band + 1 == bands;
    u32 const first_row = band * chain_band_rows_k + lane * 32u;
    u32 const my_rows = first_row >= m ? 0u : (m - first_row < 32u ? m - first_row : 32u);

    // D[i][0] = i: every vertical delta starts at +1.  Rows past the end of the pattern (last word of the last band)
    // sit BELOW the real ones and cannot influence them; they are masked out of the final count.
    u32 vp = ~0u, vn = 0;
    u32 down_bits = 0; // deltas leaving this word's last row under the K columns of the lane's latest step

    u32 *const parked_bits = parked + (u64)pair * parked_words; // [step]: written by a band, consumed by its successor, in place
    u64 *const progress_out = progress + (u64)pair * max_bands + band;
    u64 const *const progress_in = progress_out - 1;

    u32 const column_steps = (n + K - 1) / K; // steps a lane needs for the whole text
    u32 const steps = column_steps + 63;      // lane l is busy during steps [l, l + column_steps)

    // The K text bytes of step `index`, little-endian in two dwords; zero past the text.
    auto load_step_symbols = [&](u32 index, u32 &low, u32 &high) {
        low = 0, high = 0;
        u8 const *const text = reinterpret_cast<u8 const *>(candidate.address);
#pragma unroll
        for (u32 j = 0; j < K; ++j) {
            u32 const column = K * index + j;
            u32 const byte = column < n ? (u32)text[column] : 0u;
            if (j < 4) low |= byte << (8 * j);
            else high |= byte << (8 * (j - 4));
        }
    };

    // Lane 0's inputs, a chunk of 16 steps at a time, one chunk in advance (see systolic.hip): lane k < 16 holds step k's.
    u32 chunk_low = 0, chunk_high = 0, next_low, next_high, chunk_bits = 0, next_bits = 0;
    load_step_symbols(lane < chunk_steps ? lane : 0, next_low, next_high);
    u64 parked_seen = 0;
    bool abandoned = false;
    auto preload_bits = [&](u32 first_step) { // the predecessor's deltas under the steps [first_step, first_step + 16)
        u32 const last_column = K * (first_step + chunk_steps);
        u64 const needed = tag | (last_column < n ? last_column : n);
        unsigned long long wait_started = 0;
        for (u32 spins = 0; parked_seen < needed && !abandoned; ++spins) {
            parked_seen = __hip_atomic_load(progress_in, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (parked_seen >= needed) break;
            __builtin_amdgcn_s_sleep(2);
            if (spins == 0) wait_started = wall_clock64();
            bool const hopeless = (spins % 256 == 255 && wall_clock64() - wait_started > chain_patience_ticks_k) ||
                                  (spins % 1024 == 1023 && __hip_atomic_load(work_counter + 1, __ATOMIC_RELAXED,
                                                                             __HIP_MEMORY_SCOPE_AGENT) == (tag | 1));
            if (hopeless) { // never hang the device on a broken invariant: flag the call, let the host report it
                if (lane == 0) __hip_atomic_fetch_max(work_counter + 1, tag | 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                abandoned = true;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); // compiler ordering; the cells are sc1 accesses
        if (lane < chunk_steps && first_step + lane < column_steps) next_bits = chain_load(parked_bits + first_step + lane);
    };
    if (!first_band) preload_bits(0);

    // The text runs ONE STEP AHEAD of the bit-vectors: `ahead_*` of lane l are the bytes the lane consumes next step, and
    // their match masks are already on their way from LDS.
    u32 ahead_low = 0, ahead_high = 0, eq_ahead[K];
    auto advance_text = [&](u32 fed_low, u32 fed_high) {
        ahead_low = chain_from_lane_above(fed_low, ahead_low);
        ahead_high = chain_from_lane_above(fed_high, ahead_high);
#pragma unroll
        for (u32 j = 0; j < K; ++j) {
            u32 const byte = ((j < 4 ? ahead_low : ahead_high) >> (8 * (j % 4))) & 0xFFu;
            eq_ahead[j] = peq[byte * 64 + lane];
        }
    };

    auto step = [&](u32 t, u32 slot, auto predicated) {
        constexpr bool is_predicated = decltype(predicated)::value;
        u32 eq[K];
#pragma unroll
        for (u32 j = 0; j < K; ++j) eq[j] = eq_ahead[j];
        if (slot + 1 == chunk_steps)
            advance_text((u32)__builtin_amdgcn_readlane((int)next_low, 0), (u32)__builtin_amdgcn_readlane((int)next_high, 0));
        else
            advance_text((u32)__builtin_amdgcn_readlane((int)chunk_low, (int)(slot + 1)),
                         (u32)__builtin_amdgcn_readlane((int)chunk_high, (int)(slot + 1)));

        // DP row 0 grows by one per column: +1 enters the first word of the first band under every column
        u32 const fed_bits = first_band ? 0x5555u : (u32)__builtin_amdgcn_readlane((int)chunk_bits, (int)slot);
        u32 const in_bits = chain_from_lane_above(fed_bits, down_bits);
        u32 const my_step = t - lane; // wraps for lanes that have not started yet
        bool const busy = is_predicated ? my_step < column_steps : true;
        if (busy) {
            u32 out_bits = 0;
#pragma unroll
            for (u32 j = 0; j < K; ++j) {
                bool const inside = is_predicated ? K * my_step + j < n : true; // the last step may be ragged
                if (inside)
                    out_bits |= chain_word_step(vp, vn, eq[j], (in_bits >> (2 * j)) & 1u, (in_bits >> (2 * j + 1)) & 1u) << (2 * j);
            }
            down_bits = out_bits;
            if (!last_band && lane == 63) chain_store(parked_bits + my_step, out_bits);
        }
        if (!last_band && t >= 63) { // publish every 16 steps of lane 63 and at the end of the text
            u32 const parked_steps = t - 62;
            if ((parked_steps % chunk_steps == 0 || t + 1 == steps) && lane == 63) {
                u32 const parked_count = K * parked_steps < n ? K * parked_steps : n;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); // compiler ordering
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // every parked store has been acknowledged
                __hip_atomic_store(progress_out, tag | parked_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    };

    for (u32 chunk_first = 0; chunk_first < steps; chunk_first += chunk_steps) {
        if (K * chunk_first < n) { // lane 0 is about to consume the steps [chunk_first, chunk_first + 16)
            chunk_low = next_low, chunk_high = next_high, chunk_bits = next_bits;
            load_step_symbols(chunk_first + chunk_steps + (lane < chunk_steps ? lane : 0), next_low, next_high);
            if (!first_band && K * (chunk_first + chunk_steps) < n) preload_bits(chunk_first + chunk_steps);
            if (chunk_first == 0)
                advance_text((u32)__builtin_amdgcn_readlane((int)chunk_low, 0), (u32)__builtin_amdgcn_readlane((int)chunk_high, 0));
        }
        bool const steady = chunk_first >= 64 && K * (chunk_first + chunk_steps) <= n; // every lane busy with K whole columns
        if (steady) {
#pragma unroll 1
            for (u32 slot = 0; slot < chunk_steps; ++slot) step(chunk_first + slot, slot, std::false_type {});
        }
        else {
            u32 const stop = steps - chunk_first < chunk_steps ? steps - chunk_first : chunk_steps;
#pragma unroll 1
            for (u32 slot = 0; slot < stop; ++slot) step(chunk_first + slot, slot, std::true_type {});
        }
    }

    // ---- D[m][n] = D[0][n] + sum over rows of the vertical deltas at column n = n + popcount(VP) - popcount(VN)
    u32 const real_rows = my_rows >= 32 ? ~0u : ((1u << my_rows) - 1u);
    i32 const band_delta = wave_sum_i32((i32)__builtin_popcount(vp & real_rows) - (i32)__builtin_popcount(vn & real_rows));
    if (lane != 0) return;
    if (bands == 1) {
        write_result((u64)((i64)n + band_delta));
        return;
    }
    u64 *const partial_of_pair = partial + (u64)pair * max_bands;
    __hip_atomic_store(partial_of_pair + band, tag | (u32)band_delta, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the partial sum is visible before this band counts as done
    __hip_atomic_fetch_max(done + pair, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    u32 const finished = (u32)__hip_atomic_fetch_add(done + pair, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (finished + 1 != bands) return;
    i64 distance = n; // the band that finishes last adds the partial sums up
    for (u32 b = 0; b < bands; ++b)
        distance += (i32)(u32)__hip_atomic_load(partial_of_pair + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    write_result((u64)distance);
}

struct chain_layout_t {
    u64 pairs, tickets;
    u32 max_bands, parked_words, waves;
    size_t progress_at, partial_at, done_at, control_bytes, parked_bytes;
};

static chain_layout_t chain_layout(u32 queries_count, u32 candidates_count, u32 longest_query, u32 longest_candidate) {
    chain_layout_t layout;
    layout.pairs = (u64)queries_count * candidates_count;
    layout.max_bands = longest_query ? (longest_query + chain_band_rows_k - 1) / chain_band_rows_k : 1;
    // Wavefronts per workgroup.  A CU holds two 64 KB tables.  A step is one long dependency chain, so a wavefront that
    // shares its SIMD with few others advances fastest: a small batch gets 4 wavefronts per table (all of them build it,
    // whatever the candidate count) and spreads over the chip; a large one is bound by how many wavefronts advance at
    // once, and every doubling raises that - up to 8 per SIMD, 8192 on the chip (profiles/r01/chain_waves_v1.txt).
    u64 const wavefronts = layout.pairs * layout.max_bands;
    u32 waves = wavefronts <= 2048 ? 4 : wavefronts <= 4096 ? 8 : chain_max_waves_k;
    while (waves > 4 && waves / 2 >= candidates_count) waves /= 2;
    int const asked = szs_tuning_get(szs_knob_chain_waves_k); // a testing aid, like the `tier` knob (host/tuning.c)
    if (asked == 4 || asked == 8 || asked == 16) waves = (u32)asked;
    layout.waves = waves;
    layout.tickets = (u64)queries_count * ((candidates_count + waves - 1) / waves) * layout.max_bands; // workgroups
    layout.parked_words = (longest_candidate + chain_columns_k - 1) / chain_columns_k + chain_slack_words_k;
    layout.progress_at = chain_header_bytes_k; // progress and partial sums: one word per (pair, band)
    layout.partial_at = layout.progress_at + layout.pairs * layout.max_bands * sizeof(u64);
    layout.done_at = layout.partial_at + layout.pairs * layout.max_bands * sizeof(u64);
    layout.control_bytes = layout.done_at + layout.pairs * sizeof(u64);
    layout.parked_bytes = layout.pairs * layout.parked_words * sizeof(u32);
    return layout;
}

} // namespace szs_hip

extern "C" int szs_hip_myers_chain_workspace_bytes(uint32_t queries_count, uint32_t candidates_count, uint32_t longest_query,
                                                   uint32_t longest_candidate, size_t *control_bytes, size_t *parked_bytes) {
    szs_hip::chain_layout_t const layout = szs_hip::chain_layout(queries_count, candidates_count, longest_query, longest_candidate);
    if (layout.tickets * layout.waves > (1ull << 26) - 1024 || layout.pairs * layout.max_bands > (1ull << 28)) // threads < 2^32
        return 0;
    *control_bytes = layout.control_bytes, *parked_bytes = layout.parked_bytes + 256;
    return 1;
}

extern "C" int szs_hip_myers_chain(szs_string_ref_t const *queries, uint32_t queries_count, szs_string_ref_t const *candidates,
                                   uint32_t candidates_count, uint32_t longest_query, uint32_t longest_candidate,
                                   uint64_t *results, uint64_t results_row_stride, int layout_flags, void *control, void *parked,
                                   uint32_t epoch, void *stream) {
    using namespace szs_hip;
    if (!queries_count || !candidates_count) return 0;
    chain_layout_t const layout = chain_layout(queries_count, candidates_count, longest_query, longest_candidate);
    if (layout.tickets * layout.waves > (1ull << 26) - 1024) return (int)hipErrorInvalidValue;
    char *const base = static_cast<char *>(control);
    auto launch = [&](auto kernel) {
        hipLaunchKernelGGL(kernel, dim3((u32)layout.tickets), dim3(64 * layout.waves), 0, static_cast<hipStream_t>(stream), queries,
                           queries_count, candidates, candidates_count, layout.max_bands, results, results_row_stride, layout_flags,
                           reinterpret_cast<u64 *>(base), reinterpret_cast<u64 *>(base + layout.progress_at),
                           reinterpret_cast<u64 *>(base + layout.partial_at), reinterpret_cast<u64 *>(base + layout.done_at),
                           static_cast<u32 *>(parked), layout.parked_words, epoch);
    };
    switch (layout.waves) {
    case 4: launch(myers_chain_kernel<4>); break;
    case 8: launch(myers_chain_kernel<8>); break;
    default: launch(myers_chain_kernel<16>); break;
    }
    return (int)hipGetLastError();
}
