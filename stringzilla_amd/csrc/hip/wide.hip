/*
 *  wide.hip - the 64-bit cell tier: every scorer of the library on an anti-diagonal walker with int64 cells.
 *
 *  The reference widens its DP cells to 64 bits when the worst-case reach of a pair (serial.hpp:135-162: (rows + columns
 *  + tracks) x largest |cost|) leaves 32 bits, on the CPU (serial.hpp:370-386) and on the GPU
 *  (/root/reference/include/stringzillas/similarities/cuda.cuh:5568,5652,5863-5874).  Round 1 of this build refused such
 *  inputs with sz_overflow_risk_k.  They only exist at ~17-million-symbol strings (2^31 / 127), where one pair is 10^14
 *  cells: nothing about this tier is tuned for short strings, it has to be EXACT and to scale in memory, not to win a
 *  benchmark.  So it is the textbook wavefront the reference's own `diagonal walker` is (serial.hpp:778-1278):
 *
 *    - one pair per workgroup (persistent grid over the pairs of the cross-product, sized by what the workspace allows);
 *    - the DP matrix is walked anti-diagonal by anti-diagonal; the three most recent diagonals of H (and the two most recent
 *      of each gap track for Gotoh's affine recurrences) live in global memory, indexed by ROW, so a cell's three inputs are
 *      [row - 1] and [row] of the previous diagonal and [row - 1] of the one before - coalesced, no transposition;
 *    - 256 threads stride over the cells of a diagonal; one workgroup barrier per diagonal;
 *    - borders and track seeds exactly as in weighted.hip (global linear: gap x k; global affine: open + extend (k - 1),
 *      tracks = border + open + extend; local: 0 / open + extend - serial.hpp:821-823,1045-1056,1195-1201); only the
 *      substitution branch is clamped at 0 in local alignments (serial.hpp:957-965); Levenshtein engines maximise negated
 *      costs and negate the result back, like every other weighted kernel here.
 *
 *  Selected by the host when the reach rule says 32 bits could overflow, or by the `cells` knob (szs_rocm_tuning_set
 *  ("cells", "64")), which is how the tests compare it with the oracle on inputs small enough to check.
 */
#include "device_common.hpp"

namespace szs_hip {

constexpr int wide_threads_k = 256;
constexpr size_t wide_header_bytes_k = 256;

__device__ __forceinline__ i64 larger(i64 a, i64 b) { return a > b ? a : b; }

/** Diagonals a pair keeps: 3 of H, and for affine gaps 2 of each gap track. */
__host__ __device__ constexpr u32 wide_planes(bool affine) { return affine ? 7u : 3u; }

template <bool local_, bool affine_, bool uniform_, bool runes_>
__global__ __launch_bounds__(wide_threads_k) void wide_scores_kernel(
    szs_cost_model_t const *__restrict__ model, szs_string_ref_t const *__restrict__ queries, u32 queries_count,
    szs_string_ref_t const *__restrict__ candidates, u32 candidates_count, u32 longest_query, i64 *__restrict__ results,
    u64 results_row_stride, int layout, i64 *workspace, u32 *work_counter) {
    __shared__ int16_t table[32 * 32];
    __shared__ u8 class_of_byte[256];
    __shared__ u32 claimed_work;
    __shared__ i64 best_of_thread[wide_threads_k];

    if constexpr (!uniform_) {
        for (int i = threadIdx.x; i < 32 * 32; i += wide_threads_k) table[i] = model->substitution[i];
        for (int i = threadIdx.x; i < 256; i += wide_threads_k) class_of_byte[i] = model->byte_to_class[i];
    }
    i64 const gap_open = model->gap_open, gap_extend = model->gap_extend;
    i64 const uniform_match = model->uniform_match, uniform_mismatch = model->uniform_mismatch;

    size_t const plane = (size_t)longest_query + 1; // a diagonal holds one cell per row, rows 0 .. longest query
    i64 *const mine = workspace + (size_t)blockIdx.x * wide_planes(affine_) * plane;
    i64 *h[3] = {mine, mine + plane, mine + 2 * plane};
    i64 *across[2] = {mine + 3 * plane, mine + 4 * plane}; // horizontal-gap track (from the cell to the left)
    i64 *down[2] = {mine + 5 * plane, mine + 6 * plane};   // vertical-gap track (from the cell above)

    auto border = [&](u64 k) -> i64 {
        if constexpr (local_) return 0;
        if constexpr (affine_) return k ? gap_open + gap_extend * (i64)(k - 1) : 0;
        return gap_open * (i64)k;
    };

    u64 const pairs = (u64)queries_count * candidates_count;
    for (;;) {
        __syncthreads();
        if (threadIdx.x == 0) claimed_work = atomicAdd(work_counter, 1u);
        __syncthreads();
        u64 const work = claimed_work;
        if (work >= pairs) break;
        szs_string_ref_t const query = queries[work / candidates_count];
        szs_string_ref_t const candidate = candidates[work % candidates_count];
        if ((layout & SZS_LAYOUT_SYMMETRIC) && candidate.index > query.index) continue; // mirrored from below

        u64 const rows = query.length, columns = candidate.length;
        u8 const *const q_bytes = reinterpret_cast<u8 const *>(query.address);
        u8 const *const c_bytes = reinterpret_cast<u8 const *>(candidate.address);
        u32 const *const q_runes = reinterpret_cast<u32 const *>(query.address);
        u32 const *const c_runes = reinterpret_cast<u32 const *>(candidate.address);
        auto cost_of = [&](u64 row, u64 column) -> i64 { // DP cell (row, column), both 1-based
            if constexpr (runes_) return q_runes[row - 1] == c_runes[column - 1] ? uniform_match : uniform_mismatch;
            else if constexpr (uniform_) return q_bytes[row - 1] == c_bytes[column - 1] ? uniform_match : uniform_mismatch;
            else return table[(u32)class_of_byte[q_bytes[row - 1]] * 32 + class_of_byte[c_bytes[column - 1]]];
        };

        i64 best = 0;
        // Diagonal d holds the cells (row, d - row).  d = 0 is the corner, d = 1 the first border cells.
        // h[d % 3][row]; tracks [d % 2][row].  Border cells are written like scored ones.
        for (u64 d = 0; d <= rows + columns; ++d) {
            i64 *const h_now = h[d % 3];
            i64 const *const h_before = h[(d + 2) % 3], *const h_before2 = h[(d + 1) % 3];
            i64 *const across_now = across[d % 2], *const down_now = down[d % 2];
            i64 const *const across_before = across[(d + 1) % 2], *const down_before = down[(d + 1) % 2];
            u64 const first_row = d > columns ? d - columns : 0, last_row = d < rows ? d : rows;
            for (u64 row = first_row + threadIdx.x; row <= last_row; row += wide_threads_k) {
                u64 const column = d - row;
                if (row == 0 || column == 0) { // the all-gap borders and the finite "discard" seeds of the tracks
                    i64 const edge = border(row ? row : column);
                    h_now[row] = edge;
                    if constexpr (affine_) across_now[row] = down_now[row] = edge + gap_open + gap_extend;
                    continue;
                }
                i64 substituted = h_before2[row - 1] + cost_of(row, column);
                if constexpr (local_) substituted = substituted > 0 ? substituted : 0;
                i64 cell;
                if constexpr (affine_) {
                    i64 const via_across = larger(h_before[row] + gap_open, across_before[row] + gap_extend);
                    i64 const via_down = larger(h_before[row - 1] + gap_open, down_before[row - 1] + gap_extend);
                    across_now[row] = via_across, down_now[row] = via_down;
                    cell = larger(larger(via_down, via_across), substituted);
                }
                else { cell = larger(larger(h_before[row - 1], h_before[row]) + gap_open, substituted); }
                h_now[row] = cell;
                if constexpr (local_) best = cell > best ? cell : best;
            }
            __syncthreads(); // the diagonal is complete (and visible to the workgroup) before the next one reads it
        }

        i64 score;
        if constexpr (local_) {
            best_of_thread[threadIdx.x] = best;
            __syncthreads();
            score = 0;
            if (threadIdx.x == 0)
                for (int t = 0; t < wide_threads_k; ++t) score = larger(score, best_of_thread[t]);
        }
        else { score = h[(rows + columns) % 3][rows]; } // the bottom-right cell sits alone on the last diagonal
        if (threadIdx.x == 0) {
            if constexpr (uniform_) score = -score; // maximised negated costs: a distance again
            bool const transposed = (layout & SZS_LAYOUT_TRANSPOSED) != 0;
            u64 const row = transposed ? candidate.index : query.index, column = transposed ? query.index : candidate.index;
            results[row * results_row_stride + column] = score;
            if ((layout & SZS_LAYOUT_SYMMETRIC) && candidate.index != query.index) results[column * results_row_stride + row] = score;
        }
    }
}

/** Workgroups the workspace budget allows: every resident pair keeps 3 or 7 diagonals of (longest query + 1) cells. */
static u32 wide_grid(bool affine, u64 pairs, u32 longest_query) {
    size_t const per_pair = (size_t)wide_planes(affine) * ((size_t)longest_query + 1) * sizeof(i64);
    size_t const budget = (size_t)16 << 30;
    u64 resident = budget / (per_pair ? per_pair : 1);
    if (resident < 1) resident = 1;
    if (resident > 2048) resident = 2048; // 8 workgroups per CU
    return (u32)(pairs < resident ? pairs : resident);
}

} // namespace szs_hip

extern "C" size_t szs_hip_wide_workspace_bytes(int affine, uint32_t queries_count, uint32_t candidates_count,
                                               uint32_t longest_query, uint32_t longest_candidate) {
    using namespace szs_hip;
    (void)longest_candidate;
    u64 const pairs = (u64)queries_count * candidates_count;
    return wide_header_bytes_k +
           (size_t)wide_grid(affine != 0, pairs, longest_query) * wide_planes(affine != 0) * ((size_t)longest_query + 1) * sizeof(i64);
}

extern "C" int szs_hip_wide_scores(int objective, int affine, szs_cost_model_t const *model, szs_string_ref_t const *queries,
                                   uint32_t queries_count, szs_string_ref_t const *candidates, uint32_t candidates_count,
                                   uint32_t longest_query, uint32_t longest_candidate, int64_t *results,
                                   uint64_t results_row_stride, int layout, void *workspace, void *stream) {
    using namespace szs_hip;
    (void)longest_candidate;
    if (!queries_count || !candidates_count) return 0;
    u64 const pairs = (u64)queries_count * candidates_count;
    if (pairs > 0xFFFFFFF0ull) return (int)hipErrorInvalidValue; // the work counter is 32 bits wide
    hipStream_t const s = static_cast<hipStream_t>(stream);
    u32 *const counter = static_cast<u32 *>(workspace);
    hipError_t const error = hipMemsetAsync(counter, 0, sizeof(u32), s);
    if (error != hipSuccess) return (int)error;
    i64 *const diagonals = reinterpret_cast<i64 *>(static_cast<char *>(workspace) + wide_header_bytes_k);
    dim3 const grid(wide_grid(affine != 0, pairs, longest_query)), block(wide_threads_k);
#define SZS_WIDE_LAUNCH(LOCAL, AFFINE, UNIFORM, RUNES)                                                                \
    hipLaunchKernelGGL((wide_scores_kernel<LOCAL, AFFINE, UNIFORM, RUNES>), grid, block, 0, s, model, queries,        \
                       queries_count, candidates, candidates_count, longest_query, results, results_row_stride, layout, \
                       diagonals, counter)
    bool const local = objective == szs_objective_local_k || objective == szs_objective_local_saturating_k;
    bool const uniform = objective == szs_objective_distance_k || objective == szs_objective_distance_runes_k;
    bool const runes = objective == szs_objective_distance_runes_k;
    if (uniform) {
        if (runes) { if (affine) SZS_WIDE_LAUNCH(false, true, true, true); else SZS_WIDE_LAUNCH(false, false, true, true); }
        else { if (affine) SZS_WIDE_LAUNCH(false, true, true, false); else SZS_WIDE_LAUNCH(false, false, true, false); }
    }
    else if (local) { if (affine) SZS_WIDE_LAUNCH(true, true, false, false); else SZS_WIDE_LAUNCH(true, false, false, false); }
    else { if (affine) SZS_WIDE_LAUNCH(false, true, false, false); else SZS_WIDE_LAUNCH(false, false, false, false); }
#undef SZS_WIDE_LAUNCH
    return (int)hipGetLastError();
}
