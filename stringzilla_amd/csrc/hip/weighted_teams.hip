/*
 *  weighted_teams.hip - Needleman-Wunsch and Smith-Waterman scores over a class table (BLOSUM62, NUC.4.4, custom 32 x 32)
 *  with 16-bit cells: the TEAM tier.  A (pair of queries, candidate) item is scored by a team of L adjacent lanes.
 *
 *  Replaces, for the ROCm build, the reference's intra-pair tiers - one warp per pair on an anti-diagonal in shared memory
 *      score_per_cuda_warp_ / affine_score_per_cuda_warp_   /root/reference/include/stringzillas/similarities/cuda.cuh:1246-1500
 *  - and must return exactly what the reference's serial scorers return
 *      needleman_wunsch_score / smith_waterman_score        .../similarities/serial.hpp:2910-3124 (tile_scorer :778-1278).
 *
 *  Why (profiles/r02/pmc_configs.json, config 4 = 512 x 512 DNA reads of ~4 KB, Smith-Waterman, affine gaps): the lanes
 *  tier (weighted_packed.hip, one pair per lane) parks the bottom row of every 32-row strip in HBM and reads it back - 857 GB
 *  per call for 2.2 GB of strings and results - spends 5.2 VALU lane-operations per cell where the recurrence has 4.5, and
 *  ONE lane walks a 5120 x 5120 pair alone for 320 ms: the floor of any share of the batch, however many GPUs split it.
 *
 *  The shape of this tier (none of it is the reference's anti-diagonal, which moves every cell through shared memory):
 *
 *  - TEAM = L lanes of one DPP row (L = 16: the whole row).  Lane k owns strip k of a group of L strips of R query rows;
 *    it runs ONE COLUMN BEHIND lane k - 1, so what lane k - 1 produced for its bottom row in the previous step is exactly
 *    what lane k needs as its row above in this one: H, the vertical-gap track and the column's class travel from lane to
 *    lane in registers (`v_mov_b32_dpp row_shr:1`), never through memory.  Only the bottom row of the whole GROUP (L x R =
 *    512 rows) is parked for the next pass over the candidate: 1 / 16 of the lanes tier's traffic per cell, and the longest
 *    pair of a launch takes 1 / 16 of the time.
 *  - TWO QUERIES PER REGISTER (team_core.hpp): the halves of a register are the same row of two queries against the same
 *    candidate symbol, so a step's costs come from a profile keyed by ONE class - 128 bytes per class and strip - and the
 *    sixteen strips of a group fit LDS together (NUC.4.4: 34 KB, BLOSUM62: 51 KB per workgroup).
 *  - BIASED UNSIGNED CELLS (team_core.hpp): the additions of the recurrence are full-rate 32-bit `v_add_u32` on both
 *    halves at once; only the maxima are `v_pk_max_u16`.
 *  - UNPREDICATED MAIN LOOP: the skew means a team's lanes start and end at different steps.  The first L - 1 steps (fill)
 *    and the steps after the wavefront's shortest candidate ended (drain, ragged lengths) predicate every lane on its own
 *    column; everything between runs four columns per batch without a single length check, with the head lane's inputs
 *    (parked row, text dword, classes) fetched one batch ahead.  The first attempt at splitting pairs over lanes predicated
 *    every column and lost.
 *
 *  One workgroup = 256 threads = 256 / L teams = one pair of queries x 256 / L candidates; persistent grid, work items
 *  (pair of queries, candidate block) drawn heaviest first from one counter.
 */
#include "device_common.hpp"
#include "team_core.hpp"

#include <type_traits>

namespace szs_hip {

using namespace szs_team;

/** Threads of a workgroup: 256, or - teams of more than sixteen lanes - 512: their sixty-four strips of profile fill most of a CU's LDS
 *  (NUC.4.4: 132 KB), one workgroup per CU, and two wavefronts per SIMD - what the affine bodies' 256 registers allow anyway - are 512 threads. */
template <int L>
constexpr u32 team_threads() { return L > 16 ? 512u : 256u; }
constexpr u32 team_slack_columns_k = 8;      // columns the parked-row prefetch may run past the longest candidate
constexpr size_t team_header_bytes_k = 256;  // the work counter lives at the head of the workspace
constexpr size_t team_most_profile_bytes_k = 160 * 1024 - 4096; // a CU's LDS less the static arrays of the kernel

/** What lane k - 1 of the team holds in `value`; the head lane (k = 0) gets `head_value` instead. */
template <int L>
__device__ __forceinline__ u32 from_left(u32 head_value, u32 value, bool is_head) {
    if constexpr (L == 1) return head_value;
    else if constexpr (L > 16) {
        // wave_shr:1 - the whole wavefront shifts by one lane, across its four rows of sixteen (a GFX9 control; exact on gfx950 and at the
        // rate of row_shr:1: scripts/dpp_wave_probe.hip, profiles/r06/dpp_wave_probe.txt).  Lane 0 has no source and keeps `old`.
        u32 const moved = (u32)__builtin_amdgcn_update_dpp((int)head_value, (int)value, 0x138, 0xF, 0xF, false);
        if constexpr (L == 64) return moved;
        else return is_head ? head_value : moved; // lane 32 heads the second team; its source lane belongs to the first
    }
    else {
        // row_shr:1 - lane 0 of a 16-lane row has no source and keeps `old` = the head's own input
        u32 const moved = (u32)__builtin_amdgcn_update_dpp((int)head_value, (int)value, 0x111, 0xF, 0xF, false);
        if constexpr (L == 16) return moved;
        else return is_head ? head_value : moved;
    }
}

template <bool affine_>
struct parked_edge_t;
template <>
struct parked_edge_t<true> {
    u32 h, f;
};
template <>
struct parked_edge_t<false> {
    u32 h;
};

template <bool affine_>
__device__ __forceinline__ team_edge_t unpark(parked_edge_t<affine_> const &slot) {
    team_edge_t edge;
    edge.h = slot.h;
    if constexpr (affine_) edge.f = slot.f;
    else edge.f = 0;
    return edge;
}
template <bool affine_>
__device__ __forceinline__ parked_edge_t<affine_> park_of(team_edge_t const &edge) {
    parked_edge_t<affine_> slot;
    slot.h = edge.h;
    if constexpr (affine_) slot.f = edge.f;
    return slot;
}

/**
 *  @tparam local_   Smith-Waterman with gap costs <= 0 instead of Needleman-Wunsch.
 *  @tparam affine_  Gotoh's three-track recurrence instead of the single-track linear one.
 *  @tparam L        lanes per team: 1, 2, 4, 8 or 16.
 *  @tparam R        registers per track = query rows per lane and pass (a multiple of 4).
 *  @tparam W        wavefronts per SIMD the registers are allocated for.
 *  @tparam wide_    cells ordered as unsigned integers (two-input maxima, 16 bits of range) instead of as half-float
 *                   patterns (three-input maxima, 15 bits): team_core.hpp.
 *  @tparam distance_ a Levenshtein engine with non-unit UNIFORM costs: global alignment over the negated costs, the result
 *                   negated back; the profile comes from class EQUALITY (`model->uniform_match / uniform_mismatch`, negated)
 *                   instead of a 32 x 32 table, and `byte_to_class` is the batch's own dense alphabet of up to 256 bytes
 *                   (byte_presence_kernel below + host/dispatch.c).  Reference: levenshtein_distance, serial.hpp:2527-2693;
 *                   its narrow register kernels cuda.cuh:2939-3128.
 */
template <bool local_, bool affine_, bool wide_, bool distance_, int L, int R, int W>
__global__ __launch_bounds__(team_threads<L>(), W) void weighted_team_kernel(
    szs_cost_model_t const *__restrict__ model, szs_string_ref_t const *__restrict__ queries, u32 queries_count,
    szs_string_ref_t const *__restrict__ candidates, u32 candidates_count, u32 candidate_blocks, i64 *__restrict__ results,
    u64 results_row_stride, int layout_flags, char *__restrict__ parked_rows, u32 parked_columns, u32 *__restrict__ work_counter,
    u32 classes) {

    using layout = team_profile_layout<L, R>;
    using parked_t = parked_edge_t<affine_>;
    constexpr u32 team_block_threads_k = team_threads<L>();
    constexpr u32 teams = team_block_threads_k / L; // candidates per workgroup
    constexpr u32 group_rows = (u32)L * R;          // query rows per pass
    constexpr int granule_ = team_granule<local_, affine_>(); // registers a short last pass is rounded up to
    static_assert(R % 4 == 0 && (L <= 16 ? 16 % L == 0 : 64 % L == 0), "whole 16-byte profile chunks; teams inside a DPP row, or rows inside a team");

    extern __shared__ __attribute__((aligned(16))) char profile[];
    // Static LDS is kept small on purpose: BLOSUM62's sixteen strips are 51,328 bytes of profile, and THREE workgroups fit a CU
    // only while the rest stays under ~3 KB (a kilobyte more took config 3 from 14.5 to 18.2 ms).
    using class_offset_t = std::conditional_t<(255u * layout::class_bytes > 65535u), u32, unsigned short>;
    using row_class_t = std::conditional_t<distance_, unsigned short, u8>; // up to 256 classes only for uniform costs
    constexpr u32 padded_row_k = distance_ ? 0xFFFFu : 0xFFu;
    __shared__ class_offset_t class_offset_of_byte[256]; // class x layout::class_bytes: the head lanes' text -> profile row
    __shared__ row_class_t group_classes[2][group_rows]; // the classes of the group's rows, both queries; all ones: padded
    __shared__ u32 claimed_work;

    using costs_t = team_costs_t<local_, affine_, wide_, distance_>;
    costs_t const k(model->gap_open, model->gap_extend);
    int16_t const *const table = model->substitution; // [query class][candidate class], 2 KB, cache-resident
    for (u32 byte = threadIdx.x; byte < 256; byte += team_block_threads_k)
        class_offset_of_byte[byte] = (class_offset_t)((u32)model->byte_to_class[byte] * layout::class_bytes);

    u32 const lane_in_team = threadIdx.x % L, team = threadIdx.x / L;
    bool const is_head = lane_in_team == 0, is_tail = lane_in_team == L - 1;
    u32 const strip_base = layout::strip_base(lane_in_team, classes);
    // this workgroup's parked rows: [column][team], and behind them its copy of what DP row 0 hands down: [column]
    parked_t *const region = reinterpret_cast<parked_t *>(parked_rows) + (u64)blockIdx.x * parked_columns * (teams + 1);
    parked_t *const parked = region + team;
    parked_t *const border_row = region + (u64)parked_columns * teams;
    // What DP row 0 hands down is a function of the column alone: written ONCE per workgroup of the persistent grid, read by the first
    // pass of every work item with the very loads the later passes read their parked rows with (another base, stride 1 instead of
    // `teams`: both wavefront-uniform).  (Rounds 3 - 5 prefilled the parked rows with it for every work item - (longest + 9) x teams
    // stores per item: config 3, where most pairs have no other pass, wrote 1.3 GB per call for 1.1 GB of algorithmic traffic.
    // Computing it in the first pass instead - scalar arithmetic, the column is uniform - put a branch into every step of the main
    // loop and cost config 3 6.5 % (21.0 -> 19.7 TCUPS): the loop must stay one basic block.)
    for (u32 column = threadIdx.x; column < parked_columns; column += team_block_threads_k)
        border_row[column] = park_of<affine_>(team_border_edge(k, column));

    bool const transposed = (layout_flags & SZS_LAYOUT_TRANSPOSED) != 0, symmetric = (layout_flags & SZS_LAYOUT_SYMMETRIC) != 0;
    auto write_result = [&](szs_string_ref_t const &query, szs_string_ref_t const &candidate, i64 score) {
        if constexpr (distance_) score = -score; // a distance is the negated score of the negated costs
        u64 const row = transposed ? candidate.index : query.index, column = transposed ? query.index : candidate.index;
        results[row * results_row_stride + column] = score;
        if (symmetric && candidate.index != query.index) results[column * results_row_stride + row] = score;
    };

    u32 const pairs = (queries_count + 1) / 2;
    u32 const work_items = pairs * candidate_blocks;
    for (;;) {
        __syncthreads(); // the previous item's LDS (profile, claimed_work) is no longer in use
        if (threadIdx.x == 0) claimed_work = atomicAdd(work_counter, 1u);
        __syncthreads();
        u32 const work = claimed_work;
        if (work >= work_items) break;
        // candidate-block-major, heaviest block first (lev_myers.hip: myers_work_item)
        u32 const pair = work % pairs, block = candidate_blocks - 1 - work / pairs;
        szs_string_ref_t const query_low = queries[2 * pair];
        bool const has_high = 2 * pair + 1 < queries_count;
        szs_string_ref_t query_high = {0, 0, 0};
        if (has_high) query_high = queries[2 * pair + 1];
        u32 const candidate_slot = block * teams + team;
        bool const exists = candidate_slot < candidates_count;
        szs_string_ref_t candidate = {0, 0, 0};
        if (exists) candidate = candidates[candidate_slot];
        bool const live_low = exists && !(symmetric && candidate.index > query_low.index);
        bool const live_high = exists && has_high && !(symmetric && candidate.index > query_high.index);
        bool const live = live_low || live_high;
        u32 const text_length = live ? candidate.length : 0;
        // wavefront-uniform by construction; said so, they (and the step counters compared with them) live in scalar registers
        u32 const longest_in_wave = (u32)__builtin_amdgcn_readfirstlane((int)wave_max_u32(text_length));
        u32 const shortest_in_wave = (u32)__builtin_amdgcn_readfirstlane((int)~wave_max_u32(live ? ~text_length : 0u)); // over live lanes
        u32 const last_slot = (block + 1) * teams < candidates_count ? (block + 1) * teams - 1 : candidates_count - 1;
        u32 const longest_in_block = candidates[last_slot].length; // candidates ascend
        u64 const safe_address = text_length ? candidate.address : (u64)(uintptr_t)parked; // see weighted.hip
        text_stream_t text(safe_address, text_length);
        if (!text_length) text.valid_dwords = 1;
        u32 const longer = query_low.length; // queries descend

        // An empty side never enters the loop (serial.hpp:1366-1373, 1594-1605, 3077-3080).
        if (is_head) {
            if (live_low && !query_low.length) write_result(query_low, candidate, (i64)k.border(text_length));
            if (live_high && !query_high.length) write_result(query_high, candidate, (i64)k.border(text_length));
        }
        if (!longer) continue;

        (void)longest_in_block;

        u32 const passes = team_passes<L, R>(longer);
        team_rows_t<affine_, R> rows;
        u32 diagonal = 0, best[4] = {k.zero_pair, k.zero_pair, k.zero_pair, k.zero_pair};
        team_edge_t out = {0, 0};
        u32 out_row = 0;

        for (u32 pass = 0; pass < passes; ++pass) {
            u32 const first_row = pass * group_rows;
            bool const park_this_pass = pass + 1 < passes;
            // where the head lanes read what the rows above hand down: the border row (first pass) or this team's parked row
            parked_t const *const above = pass ? parked : border_row;
            u32 const above_stride = pass ? teams : 1u;

            // Rows per lane in this pass: R, or - last pass - what is left over L lanes in whole chunks of four (team_core.hpp).
            u32 const registers_now = (u32)__builtin_amdgcn_readfirstlane((int)team_pass_registers<L, R, granule_>(longer, pass));
            // whole chunks of four registers (one 16-byte read of the profile each) and, round 6, half a chunk behind them
            u32 const chunks_now = registers_now / 4, half_now = registers_now & 2u, chunks_built = (registers_now + 3u) / 4u;

            // ---- the profile of the group's L strips: thread -> (strip, class, chunk of 4 registers)
            __syncthreads(); // everyone is done with the previous pass's profile; the prefilled rows are written
            for (u32 slot = threadIdx.x; slot < 2 * (u32)L * registers_now; slot += team_block_threads_k) {
                u32 const half = slot / ((u32)L * registers_now), within = slot % ((u32)L * registers_now);
                szs_string_ref_t const &query = half ? query_high : query_low;
                u32 const row = first_row + within;
                group_classes[half][within / registers_now * R + within % registers_now] =
                    row < query.length ? (row_class_t)model->byte_to_class[reinterpret_cast<u8 const *>(query.address)[row]] : (row_class_t)padded_row_k;
            }
            __syncthreads();
            for (u32 slot = threadIdx.x; slot < (u32)L * classes * chunks_built; slot += team_block_threads_k) {
                u32 const chunk = slot % chunks_built, symbol_class = slot / chunks_built % classes, strip = slot / chunks_built / classes;
                u32 entries[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    bool const held = 4 * chunk + r < registers_now; // (the second half of a half chunk: rows nobody scores, never written above)
                    u32 const low_class = held ? (u32)group_classes[0][strip * R + 4 * chunk + r] : padded_row_k;
                    u32 const high_class = held ? (u32)group_classes[1][strip * R + 4 * chunk + r] : padded_row_k;
                    i32 low, high;
                    if constexpr (distance_) { // uniform costs: equal bytes have equal classes (serial.hpp:106-115)
                        low = low_class == padded_row_k ? 0 : low_class == symbol_class ? model->uniform_match : model->uniform_mismatch;
                        high = high_class == padded_row_k ? 0 : high_class == symbol_class ? model->uniform_match : model->uniform_mismatch;
                    }
                    else { // cost(query, candidate) = table[class(query)][class(candidate)]: the QUERY picks the row (serial.hpp:199-204)
                        low = low_class != padded_row_k ? table[low_class * 32 + symbol_class] : 0;
                        high = high_class != padded_row_k ? table[high_class * 32 + symbol_class] : 0;
                    }
                    entries[r] = k.profile_entry(low, high);
                }
                *reinterpret_cast<uint4 *>(profile + layout::strip_base(strip, classes) + symbol_class * layout::class_bytes + chunk * 16) =
                    make_uint4(entries[0], entries[1], entries[2], entries[3]);
            }
            __syncthreads();

            team_seed<costs_t, R>(k, first_row + lane_in_team * registers_now, rows, diagonal);

            // One step of this lane at DP column `column`; `head_*`: what the head lane takes instead of a neighbour's output.
            auto hand_over = [&](team_edge_t const &head_edge, u32 head_row, team_edge_t &in, u32 &in_row) {
                in.h = from_left<L>(head_edge.h, out.h, is_head);
                in.f = affine_ ? from_left<L>(head_edge.f, out.f, is_head) : 0u;
                in_row = from_left<L>(head_row, out_row, is_head);
            };
            // The profile row arrives 16 bytes at a time, one chunk ahead of the rows that consume it, and nothing moves across
            // a chunk: left alone, hipcc hoists every read of all four steps of a batch (4 R registers) above the first row.
            // The MAIN LOOP of a pass knows how many chunks the pass holds as a constant (round 4: one copy of the loop per
            // count - a pass that asks a wavefront-uniform question per chunk instead measured 14 % slower per step, and half of
            // config 3's pairs have no other pass: a 448-row protein is ONE pass of 28 of the 32 registers); the predicated
            // steps of a short pass - fill, drain - still ask, they are few.
            // `opening`: in the main loop a lane knows its row of the NEXT step when this one begins (it is its left neighbour's
            // row of this step), so the first chunk of every step is fetched a whole step early (`row_after` -> `opening`); a
            // predicated step fetches its own.
            uint4 opening = make_uint4(0, 0, 0, 0);
            auto advance = [&](auto chunks, auto pipelined, team_edge_t const &in, u32 in_row, u32 row_after) {
                // `chunks`: HALF the registers this pass holds, as a constant (2 c: c chunks of four; 2 c + 1: and half a chunk behind them) -
                // or 0: ask `chunks_now` / `half_now`, chunk by chunk
                constexpr int code_ = decltype(chunks)::value;
                constexpr int fixed_ = code_ / 2;
                constexpr bool half_ = (code_ & 1) != 0;
                constexpr bool pipelined_ = decltype(pipelined)::value;
                constexpr int most_ = code_ ? fixed_ : R / 4;
                uint4 const *const row = reinterpret_cast<uint4 const *>(profile + strip_base + in_row);
                team_step_t<costs_t, R> step;
                uint4 next = pipelined_ ? opening : row[0];
                if constexpr (pipelined_) opening = *reinterpret_cast<uint4 const *>(profile + strip_base + row_after);
                step.begin(in, diagonal, next.x);
#pragma unroll
                for (int chunk = 0; chunk < most_ + (half_ ? 1 : 0); ++chunk) {
                    // one way OUT per chunk, not a way AROUND it: the rows of a skipped chunk then need no copies to meet
                    // the rows of a scored one again (a guard around every chunk cost ten v_mov per four rows)
                    if (!code_ && (u32)chunk >= chunks_now) {
                        if (half_now) // (uniform) the pass ends on half a chunk: its two rows, and out
                            step.row(k, rows, 4 * chunk + 0, next.y, best), step.row(k, rows, 4 * chunk + 1, next.z, best);
                        break;
                    }
                    uint4 const now = next;
                    // one chunk past the last one of a short pass: inside the profile, unused.  (TWO chunks ahead measured the
                    // same within a percent - 14.27 / 525 ms against 14.10 / 529 on configs 3 / 4: it is not the LDS round trip.)
                    if (chunk + 1 < most_ + (half_ ? 1 : 0)) next = row[chunk + 1];
                    // (a row is handed the cost of the row BELOW it: team_step_t::row)
                    step.row(k, rows, 4 * chunk + 0, now.y, best), step.row(k, rows, 4 * chunk + 1, now.z, best);
                    // the half chunk behind the whole ones is a chunk like them with two rows (one body per chunk, one scheduling
                    // fence behind each: as a tail outside the loop it cost every instance 40 registers)
                    if (!(half_ && chunk == fixed_))
                        step.row(k, rows, 4 * chunk + 2, now.w, best), step.row(k, rows, 4 * chunk + 3, next.x, best);
                    __builtin_amdgcn_sched_barrier(0);
                }
                out = step.end();
                out_row = in_row;
            };
            // Predicated step `t`: the head's column is t + 1, this lane's is t + 1 - lane_in_team.  `fetched`: the head's text byte
            // and (later passes) parked entry of this column are at hand - the fill phase fetches them a batch of four steps at a time;
            // the drain phase, where few heads still have a column, fetches its own.
            auto careful_step = [&](auto chunks, u32 t, auto fetched, u32 fetched_byte, parked_t const &fetched_slot) {
                constexpr bool fetched_ = decltype(fetched)::value;
                u32 const head_column = t + 1;
                team_edge_t head_edge = {0, 0};
                u32 head_row = 0;
                if (head_column <= text_length) {
                    head_edge = unpark<affine_>(fetched_ ? fetched_slot : above[(u64)head_column * above_stride]);
                    if constexpr (fetched_) head_row = class_offset_of_byte[fetched_byte];
                    else {
                        u32 const at = text.byte_shift + head_column - 1;
                        head_row = class_offset_of_byte[(text.aligned_base[at / 4] >> (8 * (at % 4))) & 0xFFu];
                    }
                }
                team_edge_t in;
                u32 in_row;
                hand_over(head_edge, head_row, in, in_row);
                u32 const column = head_column - lane_in_team; // wraps for a lane that has not started
                if (column - 1 < text_length) {
                    advance(chunks, std::false_type {}, in, in_row, 0u);
                    if (is_tail && park_this_pass) parked[(u64)column * teams] = park_of<affine_>(out);
                }
            };

            auto walk = [&](auto chunks) {
                constexpr int code_ = decltype(chunks)::value;
                // (rounds 4 - 5: the predicated steps of a short pass asked `chunks_now` chunk by chunk.  They are instantiated per walk
                // anyway, and with half chunks the asking version grew every instance's registers: the linear 16 x 32 body 123 -> 164,
                // the affine 4 x 16 distance body to 39 spilled - so they take the walk's own constant)
                std::integral_constant<int, code_> const careful;
                constexpr u32 fill = (u32)((L - 1 + 3) / 4 * 4); // the first step at which every lane of a team has a column
                u32 t = 0;
                // ---- fill: the first lanes of every team start one after the other.  Batches of four steps: the head's four text bytes
                //      are one spliced dword, fetched - like the parked entries of a later pass - a batch ahead of the steps that use them
                //      (rounds 3 - 5: a dependent global load, an LDS lookup and - later passes - another global load INSIDE every one of
                //      these steps; config 3's candidates are ~512 columns, 16 of them walked that way per pass).  Steps past the end of
                //      a short wavefront's texts find no lane with a column and do nothing.
#ifndef SZS_TEAM_FILL_BATCHED
#define SZS_TEAM_FILL_BATCHED 1
#endif
                if (SZS_TEAM_FILL_BATCHED && fill && longest_in_wave) {
                    u32 raw_low = text.raw(0), raw_high = text.raw(1);
#pragma unroll 1
                    for (; t < fill && t < longest_in_wave + L - 1; t += 4) {
                        u32 const bytes = text.splice(raw_low, raw_high);
                        raw_low = raw_high, raw_high = text.raw(t / 4 + 2);
                        parked_t now[4] = {}; // the batch's four entries from above in flight together (one latency, not four)
#pragma unroll
                        for (u32 s = 0; s < 4; ++s)
                            if (t + 1 + s <= text_length) now[s] = above[(u64)(t + 1 + s) * above_stride];
#pragma unroll
                        for (u32 s = 0; s < 4; ++s) careful_step(careful, t + s, std::true_type {}, (bytes >> (8 * s)) & 0xFFu, now[s]);
                    }
                }
#if !SZS_TEAM_FILL_BATCHED
#pragma unroll 1
                for (; t < fill && t < longest_in_wave + L - 1; ++t) careful_step(careful, t, std::false_type {}, 0u, parked_t {});
#endif
                // ---- main loop: batches of four steps in which EVERY live lane of the wavefront has a column - no length
                //      checks, unconditional loads / stores (dead teams run along on their own parked slots).
                if (t == fill && t + 4 <= shortest_in_wave) {
                    u32 dword = t / 4; // the head's columns t + 1 ... t + 4 are the text bytes of dword t / 4
                    u32 raw_low = text.raw_clamped(dword), raw_high = text.raw_clamped(dword + 1);
                    parked_t ahead[4]; // what the head takes at the steps t ... t + 3; refilled for the next batch as it goes
#pragma unroll
                    for (int s = 0; s < 4; ++s) ahead[s] = above[(u64)(t + 1 + s) * above_stride];
                    u32 bytes_now = text.splice(raw_low, raw_high);
                    raw_low = raw_high, raw_high = text.raw_clamped(dword + 2);
                    // rows travel one step AHEAD of the cells: this lane's row of the first step of the loop ...
                    u32 my_row = from_left<L>(class_offset_of_byte[bytes_now & 0xFFu], out_row, is_head);
                    opening = *reinterpret_cast<uint4 const *>(profile + strip_base + my_row);
                    u32 head_row_after = class_offset_of_byte[(bytes_now >> 8) & 0xFFu]; // ... the head's of the step after, looked up two steps early
#pragma unroll 1
                    for (; t + 4 <= shortest_in_wave; t += 4, ++dword) {
                        // the text of the NEXT batch, one batch early (clamped reads: never past the string)
                        u32 const bytes_ahead = text.splice(raw_low, raw_high);
                        raw_low = raw_high, raw_high = text.raw_clamped(dword + 3);
#pragma unroll
                        for (int s = 0; s < 4; ++s) {
                            team_edge_t const head_edge = unpark<affine_>(ahead[s]);
                            ahead[s] = above[(u64)(t + 5 + s) * above_stride]; // the slack columns make the overrun harmless
                            u32 const my_row_after = from_left<L>(head_row_after, my_row, is_head);
                            head_row_after = class_offset_of_byte[(s < 2 ? bytes_now >> (8 * (s + 2)) : bytes_ahead >> (8 * (s - 2))) & 0xFFu];
                            team_edge_t in;
                            in.h = from_left<L>(head_edge.h, out.h, is_head);
                            in.f = affine_ ? from_left<L>(head_edge.f, out.f, is_head) : 0u;
                            advance(chunks, std::true_type {}, in, my_row, my_row_after);
                            my_row = my_row_after;
                            // the tail's column of step t + s is t + s + 2 - L >= 1: L - 1 <= fill <= t
                            if (park_this_pass && is_tail) parked[(u64)(t + s + 2 - L) * teams] = park_of<affine_>(out);
                        }
                        bytes_now = bytes_ahead;
                    }
                }
                // ---- drain: ragged lengths and the lanes that are still behind their head
#pragma unroll 1
                for (; t < longest_in_wave + L - 1; ++t) careful_step(careful, t, std::false_type {}, 0u, parked_t {});
            };
            if (longest_in_wave) {
#define SZS_TEAM_WALK(CODE) /* CODE = registers of the pass / 2 */                                                    \
    if constexpr (R / 2 >= CODE && (CODE % 2 == 0 || granule_ == 2))                                                   \
        if (registers_now == 2u * CODE) walk(std::integral_constant<int, CODE> {});
                SZS_TEAM_WALK(1) SZS_TEAM_WALK(2) SZS_TEAM_WALK(3) SZS_TEAM_WALK(4) SZS_TEAM_WALK(5) SZS_TEAM_WALK(6) SZS_TEAM_WALK(7) SZS_TEAM_WALK(8)
                SZS_TEAM_WALK(9) SZS_TEAM_WALK(10) SZS_TEAM_WALK(11) SZS_TEAM_WALK(12) SZS_TEAM_WALK(13) SZS_TEAM_WALK(14) SZS_TEAM_WALK(15) SZS_TEAM_WALK(16)
#undef SZS_TEAM_WALK
                static_assert(R / 2 <= 16, "one copy of the main loop per count of register pairs");
            }

            // ---- scores that are complete after this pass
            if constexpr (!local_) {
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    szs_string_ref_t const &query = half ? query_high : query_low;
                    if (!query.length) continue;
                    u32 last_pass, last_lane, last_reg;
                    team_last_row<L, R, granule_>(query.length, longer, last_pass, last_lane, last_reg);
                    if (pass != last_pass || lane_in_team != last_lane || !(half ? live_high : live_low)) continue;
                    u32 cell = 0;
#pragma unroll
                    for (int r = 0; r < R; ++r)
                        if ((u32)r == last_reg) cell = rows.h[r];
                    write_result(query, candidate, (i64)(k.truth(half ? (u32)high_of(cell) : (u32)low_of(cell)) - (affine_ ? 0 : k.open)));
                }
            }
        }
        if constexpr (local_) {
            u32 both_best = costs_t::order::max2(costs_t::order::max2(best[0], best[1]), costs_t::order::max2(best[2], best[3]));
#pragma unroll
            for (int offset = 1; offset < L; offset <<= 1) both_best = costs_t::order::max2(both_best, (u32)__shfl_xor((int)both_best, offset, 64));
            if (is_head) {
                if (live_low) write_result(query_low, candidate, (i64)k.truth((u32)low_of(both_best)));
                if (live_high && query_high.length) write_result(query_high, candidate, (i64)k.truth((u32)high_of(both_best)));
            }
        }
    }
}

/**
 *  Which byte values occur in a tape: 256 bits.  A Levenshtein engine's costs are uniform - only EQUALITY of symbols matters -
 *  so the host numbers the bytes that actually occur 0 ... A - 1 and the team kernel keys its profile by those classes: 95
 *  rows for printable ASCII, 4 for DNA, instead of 256.  One pass over the bytes at L2 / HBM speed, enqueued beside the planner.
 */
__global__ __launch_bounds__(256) void byte_presence_kernel(u8 const *__restrict__ data, void const *__restrict__ offsets, u32 count, u32 wide,
                                                            u32 *__restrict__ presence) {
    __shared__ u32 seen[8];
    if (threadIdx.x < 8) seen[threadIdx.x] = 0;
    __syncthreads();
    u64 const first = wide ? static_cast<u64 const *>(offsets)[0] : static_cast<u32 const *>(offsets)[0];
    u64 const last = wide ? static_cast<u64 const *>(offsets)[count] : static_cast<u32 const *>(offsets)[count];
    u32 mine[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (u64 at = first + (u64)blockIdx.x * blockDim.x + threadIdx.x; at < last; at += (u64)gridDim.x * blockDim.x) {
        u32 const byte = data[at];
#pragma unroll
        for (int word = 0; word < 8; ++word) mine[word] |= (byte >> 5) == (u32)word ? 1u << (byte & 31) : 0u;
    }
#pragma unroll
    for (int word = 0; word < 8; ++word)
        if (mine[word]) atomicOr(&seen[word], mine[word]);
    __syncthreads();
    if (threadIdx.x < 8 && seen[threadIdx.x]) atomicOr(&presence[threadIdx.x], seen[threadIdx.x]);
}

template <int L, int R>
static size_t team_profile_bytes(u32 classes) { return team_profile_layout<L, R>::total_bytes(classes); }

/** Workgroups that can be RESIDENT at once for this kernel instance with this profile size. */
template <bool local_, bool affine_, bool wide_, bool distance_, int L, int R, int W>
static u32 team_grid(u64 work_items, u32 classes) {
    static int resident_of[device_slots_k][258]; // per instance, device ordinal and class count
    int *const slot = &resident_of[device_slot()][classes];
    int resident = cached(slot);
    if (!resident) {
        int device = 0, units = 0, per_unit = 0;
        size_t const profile_bytes = team_profile_bytes<L, R>(classes);
        // the attribute is per function, not per launch: the most this instance may ever ask for, set once
        size_t const most = team_profile_bytes<L, R>(distance_ ? 256 : 32) < team_most_profile_bytes_k ? team_profile_bytes<L, R>(distance_ ? 256 : 32)
                                                                                                      : team_most_profile_bytes_k;
        constexpr u32 team_block_threads_k = team_threads<L>();
        if (hipFuncSetAttribute(reinterpret_cast<void const *>(weighted_team_kernel<local_, affine_, wide_, distance_, L, R, W>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)most) != hipSuccess ||
            hipGetDevice(&device) != hipSuccess ||
            hipDeviceGetAttribute(&units, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess ||
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_unit, weighted_team_kernel<local_, affine_, wide_, distance_, L, R, W>,
                                                         (int)team_block_threads_k, profile_bytes) != hipSuccess ||
            units <= 0 || per_unit <= 0) {
            (void)hipGetLastError();
            units = 256, per_unit = 1;
        }
        resident = units * per_unit;
        remember(slot, resident);
    }
    return (u32)(work_items < (u64)resident ? work_items : (u64)resident);
}

template <int L>
static u64 team_work_items(u32 queries_count, u32 candidates_count) {
    u32 const teams = team_threads<L>() / L;
    return (u64)((queries_count + 1) / 2) * ((candidates_count + teams - 1) / teams);
}

} // namespace szs_hip

/* The instances that are compiled: (lanes per team, registers per track, wavefronts per SIMD), each in both orders. */
#ifndef SZS_TEAM_SHAPES
#define SZS_TEAM_SHAPES(CALL) CALL(16, 32, 2) CALL(16, 16, 4) CALL(4, 32, 2) CALL(4, 16, 4) CALL(64, 32, 2)
#endif

/* `objective`: 0 global (Needleman-Wunsch), 1 local (Smith-Waterman, gaps <= 0), 2 distance (Levenshtein, uniform costs). */
#define SZS_TEAM_DISPATCH(L, R, W, CALL)                                                                               \
    if (shape == L * 10000u + R * 100u + W) {                                                                          \
        if (objective == 2 && affine && wide) CALL(false, true, true, true, L, R, W);                                  \
        if (objective == 2 && affine) CALL(false, true, false, true, L, R, W);                                         \
        if (objective == 2 && wide) CALL(false, false, true, true, L, R, W);                                           \
        if (objective == 2) CALL(false, false, false, true, L, R, W);                                                  \
        if (objective == 1 && affine && wide) CALL(true, true, true, false, L, R, W);                                  \
        if (objective == 1 && affine) CALL(true, true, false, false, L, R, W);                                         \
        if (objective == 1 && wide) CALL(true, false, true, false, L, R, W);                                           \
        if (objective == 1) CALL(true, false, false, false, L, R, W);                                                  \
        if (affine && wide) CALL(false, true, true, false, L, R, W);                                                   \
        if (affine) CALL(false, true, false, false, L, R, W);                                                          \
        if (wide) CALL(false, false, true, false, L, R, W);                                                            \
        CALL(false, false, false, false, L, R, W);                                                                     \
    }

extern "C" unsigned szs_hip_weighted_team_shape(unsigned index) {
    unsigned const shapes[] = {
#define SZS_TEAM_SHAPE_CODE(L, R, W) L * 10000u + R * 100u + W,
        SZS_TEAM_SHAPES(SZS_TEAM_SHAPE_CODE)
#undef SZS_TEAM_SHAPE_CODE
    };
    return index < sizeof(shapes) / sizeof(shapes[0]) ? shapes[index] : 0;
}

extern "C" int szs_hip_weighted_team_has_shape(unsigned shape) {
    for (unsigned index = 0; szs_hip_weighted_team_shape(index); ++index)
        if (szs_hip_weighted_team_shape(index) == shape) return 1;
    return 0;
}

/** Candidates per work item (= teams per workgroup) of a shape; 0: not compiled. */
extern "C" unsigned szs_hip_weighted_team_candidates_per_item(unsigned shape) {
    using namespace szs_hip;
#define SZS_TEAM_SHAPE_TEAMS(L, R, W)                                                                                  \
    if (shape == L * 10000u + R * 100u + W) return team_threads<L>() / L;
    SZS_TEAM_SHAPES(SZS_TEAM_SHAPE_TEAMS)
#undef SZS_TEAM_SHAPE_TEAMS
    return 0;
}

extern "C" uint32_t szs_hip_weighted_team_reach_limit(int objective, int wide) { return szs_team::team_reach_limit(objective, wide != 0); }

/** Does the profile of `classes` classes fit a CU's LDS for this shape?  (Sixteen lanes x 32 registers: up to 77 classes.) */
extern "C" int szs_hip_weighted_team_fits(unsigned shape, uint32_t classes) {
    using namespace szs_hip;
#define SZS_TEAM_SHAPE_FITS(L, R, W)                                                                                   \
    if (shape == L * 10000u + R * 100u + W) return team_profile_bytes<L, R>(classes) <= team_most_profile_bytes_k;
    SZS_TEAM_SHAPES(SZS_TEAM_SHAPE_FITS)
#undef SZS_TEAM_SHAPE_FITS
    return 0;
}

extern "C" size_t szs_hip_weighted_team_workspace_bytes(int objective, int affine, int wide, unsigned shape, uint32_t classes,
                                                        uint32_t queries_count, uint32_t candidates_count, uint32_t longest_candidate) {
    using namespace szs_hip;
    if (classes > (objective == 2 ? 256u : 32u) || !szs_hip_weighted_team_fits(shape, classes)) return 0;
#define SZS_TEAM_BYTES(LOCAL, AFFINE, WIDE, DISTANCE, L, R, W)                                                         \
    return team_header_bytes_k + (size_t)team_grid<LOCAL, AFFINE, WIDE, DISTANCE, L, R, W>(team_work_items<L>(queries_count, candidates_count), classes) * \
                                     (longest_candidate + 1 + team_slack_columns_k) * (team_threads<L>() / L + 1) * \
                                     sizeof(parked_edge_t<AFFINE>)
#define SZS_TEAM_SHAPE_BYTES(L, R, W) SZS_TEAM_DISPATCH(L, R, W, SZS_TEAM_BYTES)
    SZS_TEAM_SHAPES(SZS_TEAM_SHAPE_BYTES)
#undef SZS_TEAM_SHAPE_BYTES
#undef SZS_TEAM_BYTES
    return 0;
}

extern "C" int szs_hip_weighted_team_scores(int objective, int affine, int wide, unsigned shape, uint32_t classes, szs_cost_model_t const *model,
                                            szs_string_ref_t const *queries, uint32_t queries_count, szs_string_ref_t const *candidates,
                                            uint32_t candidates_count, uint32_t longest_candidate, int64_t *results,
                                            uint64_t results_row_stride, int layout_flags, void *workspace, void *stream) {
    using namespace szs_hip;
    if (!queries_count || !candidates_count) return 0;
    if (classes > (objective == 2 ? 256u : 32u) || !szs_hip_weighted_team_fits(shape, classes)) return (int)hipErrorInvalidValue;
    u32 *const counter = static_cast<u32 *>(workspace);
    char *const parked = static_cast<char *>(workspace) + team_header_bytes_k;
    hipStream_t const s = static_cast<hipStream_t>(stream);
#define SZS_TEAM_LAUNCH(LOCAL, AFFINE, WIDE, DISTANCE, L, R, W)                                                                  \
    {                                                                                                                  \
        u64 const work_items = team_work_items<L>(queries_count, candidates_count);                                    \
        if (work_items > 0xFFFFFFF0ull) return (int)hipErrorInvalidValue; /* the host cuts larger cross-products */     \
        hipError_t const error = hipMemsetAsync(counter, 0, sizeof(u32), s);                                           \
        if (error != hipSuccess) return (int)error;                                                                    \
        u32 const grid = team_grid<LOCAL, AFFINE, WIDE, DISTANCE, L, R, W>(work_items, classes);                                 \
        u32 const teams = team_threads<L>() / L;                                                                       \
        size_t const profile_bytes = team_profile_bytes<L, R>(classes);                                                \
        hipLaunchKernelGGL((weighted_team_kernel<LOCAL, AFFINE, WIDE, DISTANCE, L, R, W>), dim3(grid), dim3(team_threads<L>()), \
                           profile_bytes, s, model, queries, queries_count, candidates,                                \
                           candidates_count, (candidates_count + teams - 1) / teams, results, results_row_stride,      \
                           layout_flags, parked, longest_candidate + 1 + team_slack_columns_k, counter, classes);      \
        return (int)hipGetLastError();                                                                                 \
    }
#define SZS_TEAM_SHAPE_LAUNCH(L, R, W) SZS_TEAM_DISPATCH(L, R, W, SZS_TEAM_LAUNCH)
    SZS_TEAM_SHAPES(SZS_TEAM_SHAPE_LAUNCH)
#undef SZS_TEAM_SHAPE_LAUNCH
#undef SZS_TEAM_LAUNCH
    return (int)hipErrorInvalidValue;
}

extern "C" int szs_hip_byte_presence(void const *data, void const *offsets, uint32_t count, int wide, uint32_t *presence, void *stream) {
    using namespace szs_hip;
    if (!count) return 0;
    hipLaunchKernelGGL(byte_presence_kernel, dim3(512), dim3(256), 0, static_cast<hipStream_t>(stream), static_cast<u8 const *>(data), offsets,
                       count, (u32)(wide != 0), presence);
    return (int)hipGetLastError();
}
