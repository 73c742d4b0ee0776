/*
 *  weighted.hip - Needleman-Wunsch, Smith-Waterman and non-unit-cost Levenshtein scores on gfx950, linear and
 *  affine (Gotoh) gaps, class-table or uniform substitution costs.
 *
 *  Replaces, for the ROCm build, the reference's weighted tiers
 *      weighted_needleman/gotoh_per_cuda_thread_, (affine_)score_per_cuda_warp_, (affine_)score_across_cuda_device_
 *      /root/reference/include/stringzillas/similarities/cuda.cuh:729-1500,5386-5760
 *  and must return exactly what the reference's serial scorers return
 *      needleman_wunsch_score / smith_waterman_score / levenshtein_distance
 *      .../similarities/serial.hpp:2527-2693,2910-3124   (recurrences: tile_scorer, serial.hpp:778-1278).
 *
 *  MI355X-first design - inter-sequence, register-striped, no cross-lane traffic:
 *
 *  - One workgroup = one QUERY x 256 CANDIDATES, one (query, candidate) pair per lane.  Lanes never talk to each
 *    other: there is no anti-diagonal, no shuffle, no shared DP row.  Utilisation comes from the batch (the C-ABI is
 *    a cross-product), and candidates are length-sorted so a wavefront's 64 texts end together.
 *  - The DP matrix of a pair is walked in horizontal STRIPS of `rows_` query rows.  Inside a strip the lane sweeps
 *    the candidate left to right; the strip's column of `rows_` cells (and its horizontal-gap track for affine gaps)
 *    lives in VGPRs.  Only the strip's bottom row must survive until the next strip: it is parked in a per-lane
 *    BOUNDARY array in global memory laid out [column][lane], so a wavefront reads/writes one coalesced 256-byte
 *    line per column.  Traffic: 2 x 4 B per `rows_` cells (x2 for affine).
 *  - Substitution costs come from a per-strip QUERY PROFILE in LDS: profile[candidate class][row] = cost(query[row],
 *    class) as packed int8, so ONE ds_read_b128 hands a lane the costs of 16 cells.  The query is shared by the whole
 *    workgroup, so the profile is built once per strip by 32 (class table) or 256 (uniform costs) threads.
 *  - Cells are 32-bit.  The host refuses inputs whose worst-case reach (serial.hpp:135-162) leaves int32.
 *  - Persistent grid: workgroups stride over (query, candidate-block) work items, so the boundary workspace is sized
 *    by the number of RESIDENT workgroups, not by the size of the results matrix.
 *
 *  Exact boundary values (parity traps of SURVEY.md section 8a) are spelled out next to the code that uses them.
 */
#include "device_common.hpp"

namespace szs_hip {

constexpr int weighted_rows_k = 16;          // strip height: 16 int8 costs = one ds_read_b128
constexpr u32 weighted_block_threads_k = 256;
constexpr u32 weighted_max_resident_blocks_k = 256 * 4; // persistent grid ceiling: 256 CUs x 4 workgroups

__device__ __forceinline__ i32 max2(i32 a, i32 b) { return a > b ? a : b; }
__device__ __forceinline__ i32 max3(i32 a, i32 b, i32 c) { return max2(max2(a, b), c); }
__device__ __forceinline__ i32 cost_byte(uint4 const &packed, int row) {
    u32 const word = row < 4 ? packed.x : row < 8 ? packed.y : row < 12 ? packed.z : packed.w;
    return (i32)(int8_t)(word >> (8 * (row & 3)));
}

/**
 *  @tparam local_    Smith-Waterman (best cell, substitution branch clamped at 0) instead of a global alignment.
 *  @tparam affine_   Gotoh's three-track recurrence instead of the single-track linear one.
 *  @tparam uniform_  costs are (match, mismatch) on raw bytes - weighted Levenshtein, computed as a maximisation of
 *                    negated costs and negated back on output - instead of the 32x32 class table.
 */
template <bool local_, bool affine_, bool uniform_>
__global__ __launch_bounds__(256) void weighted_scores_kernel(
    szs_cost_model_t const *__restrict__ model, szs_string_ref_t const *__restrict__ queries, u32 queries_count,
    szs_string_ref_t const *__restrict__ candidates, u32 candidates_count, u32 candidate_blocks,
    i64 *__restrict__ results, u64 results_row_stride, int symmetric, i32 *__restrict__ boundary, u32 boundary_columns) {

    constexpr int rows = weighted_rows_k;
    constexpr int classes = uniform_ ? 256 : 32;
    __shared__ __attribute__((aligned(16))) int8_t profile[classes * rows];
    __shared__ u8 class_of_byte[256];

    i32 const gap_open = model->gap_open, gap_extend = model->gap_extend;
    if constexpr (!uniform_) class_of_byte[threadIdx.x] = model->byte_to_class[threadIdx.x];

    // This workgroup's private boundary rows: [column][lane], one plane for H and one for the vertical-gap track.
    u64 const plane = (u64)boundary_columns * weighted_block_threads_k;
    i32 *const boundary_scores = boundary + (u64)blockIdx.x * plane * (affine_ ? 2 : 1) + threadIdx.x;
    i32 *const boundary_gaps = boundary_scores + plane;

    u64 const work_items = (u64)queries_count * candidate_blocks;
    for (u64 work = blockIdx.x; work < work_items; work += gridDim.x) {
        szs_string_ref_t const query = queries[work / candidate_blocks];
        u32 const candidate_slot = (u32)(work % candidate_blocks) * weighted_block_threads_k + threadIdx.x;
        bool live = candidate_slot < candidates_count;
        szs_string_ref_t candidate = {0, 0, 0};
        if (live) candidate = candidates[candidate_slot];
        if (symmetric && candidate.index > query.index) live = false;
        u32 const text_length = live ? candidate.length : 0;
        u32 const longest_in_wave = wave_max_u32(text_length);
        text_stream_t const text(candidate.address, text_length);
        u8 const *const pattern = reinterpret_cast<u8 const *>(query.address);
        u32 const query_length = query.length;

        // Value of DP cell (row `i`, column 0) and (row 0, column `j`): the all-gap borders.
        //   global linear : gap * k                                   (serial.hpp:821-823)
        //   global affine : k ? open + extend * (k - 1) : 0           (serial.hpp:1045-1047)
        //   local         : 0
        auto border = [&](u32 k) -> i32 {
            if constexpr (local_) return 0;
            if constexpr (affine_) return k ? gap_open + gap_extend * (i32)(k - 1) : 0;
            return gap_open * (i32)k;
        };

        // A pair with an empty side never enters the column loop: its score is the border itself (local: 0).
        i32 score = local_ ? 0 : border(query_length ? query_length : text_length);

        for (u32 first_row = 0; first_row < query_length; first_row += rows) {
            u32 const rows_here = query_length - first_row < (u32)rows ? query_length - first_row : (u32)rows;
            bool const is_first_strip = first_row == 0;
            bool const is_last_strip = first_row + rows >= query_length;

            // ---- query profile of this strip
            __syncthreads(); // everyone is done with the previous strip's profile (and class_of_byte is written)
            if (threadIdx.x < (u32)classes) {
                u32 packed[rows / 4] = {0, 0, 0, 0};
#pragma unroll
                for (int r = 0; r < rows; ++r) {
                    i32 cost = 0;
                    if ((u32)r < rows_here) {
                        u8 const symbol = pattern[first_row + r];
                        if constexpr (uniform_)
                            cost = symbol == threadIdx.x ? model->uniform_match : model->uniform_mismatch;
                        else // cost(query, candidate) = table[class(query)][class(candidate)]: the QUERY picks the row
                            cost = model->substitution[(u32)class_of_byte[symbol] * 32 + threadIdx.x];
                    }
                    packed[r / 4] |= ((u32)cost & 0xFFu) << (8 * (r % 4));
                }
                reinterpret_cast<uint4 *>(profile)[threadIdx.x] = make_uint4(packed[0], packed[1], packed[2], packed[3]);
            }
            __syncthreads();

            // ---- column 0 of the strip.  Track seeds are the FINITE "discard" values of the reference:
            //      global: border + open + extend (serial.hpp:1049-1056); local: open + extend (serial.hpp:1195-1201).
            i32 cells[rows], gaps_across[affine_ ? rows : 1];
#pragma unroll
            for (int r = 0; r < rows; ++r) {
                cells[r] = border(first_row + r + 1);
                if constexpr (affine_) gaps_across[r] = cells[r] + gap_open + gap_extend;
            }
            i32 above_left = border(first_row); // DP cell (first_row, column - 1)

            u32 raw_low = text.raw(0), raw_high = text.raw(1);
            for (u32 column = 0; column < longest_in_wave; column += 4) {
                u32 const symbols = text.splice(raw_low, raw_high);
                raw_low = raw_high;
                raw_high = text.raw(column / 4 + 2);
#pragma unroll
                for (int step = 0; step < 4; ++step) {
                    u32 const j = column + step + 1; // 1-based DP column
                    if (j > text_length) continue;   // this lane's text has ended; others in the wave go on
                    u32 const symbol = (symbols >> (8 * step)) & 0xFFu;
                    u32 const klass = uniform_ ? symbol : (u32)class_of_byte[symbol];
                    uint4 const costs = reinterpret_cast<uint4 const *>(profile)[klass];

                    // The row above the strip at this column: the border for the first strip, else the parked boundary.
                    i32 above, above_gap = 0;
                    if (is_first_strip) {
                        above = border(j);
                        if constexpr (affine_) above_gap = above + gap_open + gap_extend;
                    }
                    else {
                        above = boundary_scores[(u64)j * weighted_block_threads_k];
                        if constexpr (affine_) above_gap = boundary_gaps[(u64)j * weighted_block_threads_k];
                    }

                    i32 diagonal = above_left;
                    above_left = above;
#pragma unroll
                    for (int r = 0; r < rows; ++r) {
                        i32 const left = cells[r];
                        i32 substituted = diagonal + cost_byte(costs, r);
                        if constexpr (local_) substituted = max2(substituted, 0); // only this branch is clamped
                        i32 cell;
                        if constexpr (affine_) {
                            i32 const gap_across = max2(left + gap_open, gaps_across[r] + gap_extend);
                            i32 const gap_down = max2(above + gap_open, above_gap + gap_extend);
                            cell = max3(gap_down, gap_across, substituted);
                            gaps_across[r] = gap_across;
                            above_gap = gap_down;
                        }
                        else { cell = max2(max2(above, left) + gap_open, substituted); }
                        if constexpr (local_) {
                            if ((u32)r < rows_here) score = max2(score, cell); // padded rows never count
                        }
                        diagonal = left;
                        above = cell;
                        cells[r] = cell;
                    }
                    if (!is_last_strip) {
                        boundary_scores[(u64)j * weighted_block_threads_k] = cells[rows - 1];
                        if constexpr (affine_) boundary_gaps[(u64)j * weighted_block_threads_k] = above_gap;
                    }
                    else if constexpr (!local_) {
                        if (j == text_length) { // bottom-right cell: last real row of the last strip
#pragma unroll
                            for (int r = 0; r < rows; ++r)
                                if ((u32)r + 1 == rows_here) score = cells[r];
                        }
                    }
                }
            }
        }

        if (live) {
            i64 const value = uniform_ ? -(i64)score : (i64)score;
            results[(u64)query.index * results_row_stride + candidate.index] = value;
            if (symmetric && candidate.index != query.index)
                results[(u64)candidate.index * results_row_stride + query.index] = value;
        }
    }
}

static u32 weighted_grid(u32 queries_count, u32 candidates_count) {
    u64 const blocks = (candidates_count + weighted_block_threads_k - 1) / weighted_block_threads_k;
    u64 const work = (u64)queries_count * blocks;
    return (u32)(work < weighted_max_resident_blocks_k ? work : weighted_max_resident_blocks_k);
}

template <bool local_, bool affine_, bool uniform_>
static int launch_weighted(szs_cost_model_t const *model, szs_string_ref_t const *queries, u32 queries_count,
                           szs_string_ref_t const *candidates, u32 candidates_count, u32 longest_candidate, i64 *results,
                           u64 stride, int symmetric, void *boundary, hipStream_t stream) {
    u32 const candidate_blocks = (candidates_count + weighted_block_threads_k - 1) / weighted_block_threads_k;
    u32 const grid = weighted_grid(queries_count, candidates_count);
    hipLaunchKernelGGL((weighted_scores_kernel<local_, affine_, uniform_>), dim3(grid), dim3(weighted_block_threads_k), 0,
                       stream, model, queries, queries_count, candidates, candidates_count, candidate_blocks, results,
                       stride, symmetric, static_cast<i32 *>(boundary), longest_candidate + 1);
    return (int)hipGetLastError();
}

} // namespace szs_hip

extern "C" size_t szs_hip_weighted_boundary_bytes(int affine, uint32_t queries_count, uint32_t candidates_count,
                                                  uint32_t longest_candidate) {
    using namespace szs_hip;
    return (size_t)weighted_grid(queries_count, candidates_count) * (longest_candidate + 1) * weighted_block_threads_k *
           sizeof(i32) * (affine ? 2 : 1);
}

extern "C" int szs_hip_weighted_scores(int objective, int affine, szs_cost_model_t const *model,
                                       szs_string_ref_t const *queries, uint32_t queries_count,
                                       szs_string_ref_t const *candidates, uint32_t candidates_count,
                                       uint32_t longest_candidate, int64_t *results, uint64_t results_row_stride,
                                       int symmetric, void *boundary, void *stream) {
    using namespace szs_hip;
    if (!queries_count || !candidates_count) return 0;
    hipStream_t const s = static_cast<hipStream_t>(stream);
#define SZS_WEIGHTED(LOCAL, AFFINE, UNIFORM)                                                                           \
    return launch_weighted<LOCAL, AFFINE, UNIFORM>(model, queries, queries_count, candidates, candidates_count,       \
                                                   longest_candidate, results, results_row_stride, symmetric, boundary, s)
    switch (objective) {
    case szs_objective_global_k:
        if (affine) SZS_WEIGHTED(false, true, false);
        SZS_WEIGHTED(false, false, false);
    case szs_objective_local_k:
        if (affine) SZS_WEIGHTED(true, true, false);
        SZS_WEIGHTED(true, false, false);
    case szs_objective_distance_k:
        if (affine) SZS_WEIGHTED(false, true, true);
        SZS_WEIGHTED(false, false, true);
    default: return (int)hipErrorInvalidValue;
    }
#undef SZS_WEIGHTED
}
