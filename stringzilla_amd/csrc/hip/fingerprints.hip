/*
 *  fingerprints.hip - rolling MinHash / Count-Min fingerprints on gfx950.
 *
 *  Replaces, for the ROCm build, the reference's fingerprint kernels
 *      /root/reference/include/stringzillas/fingerprints/cuda.cuh:102,233,409
 *  and must return exactly what its serial engines return
 *      floating_rolling_hashers<sz_cap_serial_k, 64>::fingerprint_chunk   .../fingerprints/serial.hpp:1212-1278
 *      basic_rolling_hashers<floating_rolling_hasher<f64_t>, u32_t>        .../fingerprints/serial.hpp:780-860
 *  (both yield, per dimension, the minimum over all windows of the polynomial hash
 *   sum (byte_i + 1) multiplier^(width-1-i)  mod  modulo,   its low 32 bits, and how often that minimum occurs).
 *
 *  Every value is an integer below 2^52 carried exactly in a double, like in the reference, so the hash of a window is a
 *  canonical residue that does not depend on HOW it was reached.  That freedom is used twice:
 *
 *  - one fused update per byte: x = state * multiplier + (new + 1) + complement * (old + 1), complement = (-multiplier^width)
 *    mod modulo >= 0, bounded by ~896 modulo < 2^52 (the reference's own single-reduction `roll`, serial.hpp:536-546),
 *    reduced by ONE Barrett step whose reciprocal is rounded DOWN, so the quotient never overshoots and a single
 *    conditional subtraction lands in [0, modulo) - 9 f64 ops instead of the 16 of the two-reduction form;
 *  - a text is cut into SEGMENTS of 4096 window positions that are hashed independently (each warms up on the width - 1
 *    bytes before it) and merged afterwards - min of the minima, sum of the counts that belong to it - so a single
 *    100 KB document occupies 25 x dims / 64 wavefronts instead of dims / 64.
 *
 *  Mapping: lane = one dimension, wavefront = 64 consecutive dimensions (the reference's "slice": one window width per
 *  wavefront on its fast path, so control flow is uniform), workgroup = 256 dimensions x one segment.  The segment's
 *  bytes are staged once in LDS and read back as broadcasts (`ds_read_u8`, every lane the same address).
 *  HBM traffic: each text byte once per 256 dimensions; 8 bytes out per (text, dimension).
 */
#include "device_common.hpp"

namespace szs_hip {

constexpr u32 fingerprint_segment_k = SZS_FINGERPRINT_SEGMENT;  // window END positions per segment
constexpr u32 fingerprint_max_width_k = SZS_FINGERPRINT_MAX_WIDTH;
constexpr u32 fingerprint_threads_k = 256;
constexpr double fingerprint_skipped_k = 1.7976931348623157e308; // no window yet: above every hash (serial.hpp:1133)

struct fingerprint_state_t {
    double minimum;
    u32 count;
};

/** Folds `other` (a later stretch of the same text) into `into`. */
__device__ __forceinline__ void fingerprint_merge(fingerprint_state_t &into, fingerprint_state_t const &other) {
    if (other.minimum < into.minimum) into = other;
    else if (other.minimum == into.minimum) into.count += other.count;
}

__device__ __forceinline__ void fingerprint_export(fingerprint_state_t const &state, u32 *hash_out, u32 *count_out) {
    bool const skipped = state.minimum == fingerprint_skipped_k; // the text is shorter than the window (serial.hpp:1188-1192)
    *hash_out = skipped ? 0xFFFFFFFFu : (u32)((u64)state.minimum & 0xFFFFFFFFull);
    *count_out = skipped ? 0u : state.count;
}

/**
 *  One workgroup: 256 dimensions (blockIdx.y) of one segment (blockIdx.x) of one text.
 *
 *  @param segment_text     [segments] the text slot every segment belongs to, or NULL when every text has one segment
 *  @param segment_prefix   [texts + 1] global index of each text's first segment
 *  @param partial_prefix   [texts + 1] index of each text's first PARTIAL slot; texts of one segment have none and write
 *                          their fingerprint straight to the outputs (the common case: many short documents)
 *  @param widths, multipliers, modulos, reciprocals, complements   [dimensions] per-dimension parameters (host/fingerprint_engines.c)
 */
template <bool staged_>
__global__ __launch_bounds__(fingerprint_threads_k) void fingerprint_segments_kernel(
    szs_string_ref_t const *__restrict__ texts, u32 first_segment, u32 const *__restrict__ segment_text,
    u32 const *__restrict__ segment_prefix, u32 const *__restrict__ partial_prefix, u32 dimensions,
    u32 const *__restrict__ widths,
    double const *__restrict__ multipliers, double const *__restrict__ modulos, double const *__restrict__ reciprocals,
    double const *__restrict__ complements, double *__restrict__ partial_minimums, u32 *__restrict__ partial_counts,
    u32 *__restrict__ min_hashes, u64 min_hashes_stride, u32 *__restrict__ min_counts, u64 min_counts_stride) {

    // `staged_`: the segment's bytes and the widest window's warm-up in LDS (every width up to 1024); windows wider than that
    // (round 4: up to SZS_FINGERPRINT_WIDEST) read the text where it lies - every lane of a wavefront the same byte, one line
    // per read, L2-resident.
    __shared__ u8 staged[staged_ ? fingerprint_segment_k + fingerprint_max_width_k : 16];

    // Which text does this segment belong to?  One segment per text (every text shorter than 4096 bytes, the common
    // case) needs no table; otherwise the host lists the owner of every segment.
    u32 const segment = first_segment + blockIdx.x;
    u32 const text_slot = segment_text ? segment_text[segment] : segment;
    szs_string_ref_t const text = texts[text_slot];
    u32 const segments_of_text = segment_prefix[text_slot + 1] - segment_prefix[text_slot];
    u32 const local_segment = segment - segment_prefix[text_slot];

    u32 const dimension = blockIdx.y * fingerprint_threads_k + threadIdx.x;
    bool const owns_dimension = dimension < dimensions;
    u32 const width = owns_dimension ? widths[dimension] : 2u;
    double const multiplier = owns_dimension ? multipliers[dimension] : 0.0;
    double const modulo = owns_dimension ? modulos[dimension] : 1.0;
    double const reciprocal = owns_dimension ? reciprocals[dimension] : 0.0;
    double const complement = owns_dimension ? complements[dimension] : 0.0;

    // Window END positions of this segment: [first_end, last_end) clipped to the text; staged bytes start `max width - 1`
    // before them so that every dimension finds its warm-up bytes.
    u32 const first_end = local_segment * fingerprint_segment_k;
    u32 const last_end = first_end + fingerprint_segment_k < text.length ? first_end + fingerprint_segment_k : text.length;
    u32 const staged_from = first_end >= fingerprint_max_width_k - 1 ? first_end - (fingerprint_max_width_k - 1) : 0u;
    u8 const *const bytes = reinterpret_cast<u8 const *>(text.address);
    if constexpr (staged_) {
        for (u32 i = staged_from + threadIdx.x; i < last_end; i += fingerprint_threads_k) staged[i - staged_from] = bytes[i];
        __syncthreads();
    }

    fingerprint_state_t state = {fingerprint_skipped_k, 0};
    // This dimension's windows end at positions >= width - 1.
    u32 const my_first_end = first_end > width - 1 ? first_end : width - 1;
    if (owns_dimension && my_first_end < last_end) {
        u8 const *const window_bytes = staged_ ? staged - staged_from : bytes; // window_bytes[i] = byte i of the text
        auto term = [&](u32 i) -> double { return (double)((u32)window_bytes[i] + 1u); };
        // x < 2^52 -> x mod modulo: Barrett with a reciprocal rounded DOWN, so the quotient is floor(x / modulo) or one
        // less and a single conditional subtraction finishes the job
        auto reduce = [&](double x) -> double {
            double const quotient = __builtin_floor(x * reciprocal);
            double const residue = __builtin_fma(-quotient, modulo, x);
            return residue >= modulo ? residue - modulo : residue;
        };
        // ---- the first window of this stretch: `width` bytes enter, none leaves
        u32 i = my_first_end - (width - 1);
        double hash = 0.0;
        for (; i <= my_first_end; ++i) hash = reduce(__builtin_fma(hash, multiplier, term(i)));
        state.minimum = hash, state.count = 1;
        // ---- every further position: one byte enters, one leaves, one window ends - straight-line, unrolled so that the
        // LDS byte reads of the next positions are in flight under the arithmetic of the current one
#pragma unroll 4
        for (; i < last_end; ++i) {
            double x = __builtin_fma(hash, multiplier, term(i));
            x = __builtin_fma(complement, term(i - width), x);
            hash = reduce(x);
            bool const smaller = hash < state.minimum;
            state.count = smaller ? 1u : state.count + (hash == state.minimum ? 1u : 0u);
            state.minimum = smaller ? hash : state.minimum;
        }
    }
    if (!owns_dimension) return;

    if (segments_of_text == 1) { // the whole text: export the fingerprint itself
        u32 *const hash_row = reinterpret_cast<u32 *>(reinterpret_cast<char *>(min_hashes) + (u64)text.index * min_hashes_stride);
        u32 *const count_row = reinterpret_cast<u32 *>(reinterpret_cast<char *>(min_counts) + (u64)text.index * min_counts_stride);
        fingerprint_export(state, hash_row + dimension, count_row + dimension);
    }
    else {
        u64 const slot = ((u64)partial_prefix[text_slot] + local_segment) * dimensions + dimension;
        partial_minimums[slot] = state.minimum, partial_counts[slot] = state.count;
    }
}

/** One thread per (multi-segment text, dimension): folds the text's partial states in text order and exports. */
__global__ __launch_bounds__(fingerprint_threads_k) void fingerprint_merge_kernel(
    szs_string_ref_t const *__restrict__ texts, u32 const *__restrict__ merge_list, u32 merge_count,
    u32 const *__restrict__ segment_prefix, u32 const *__restrict__ partial_prefix, u32 dimensions,
    double const *__restrict__ partial_minimums, u32 const *__restrict__ partial_counts, u32 *__restrict__ min_hashes,
    u64 min_hashes_stride, u32 *__restrict__ min_counts, u64 min_counts_stride) {
    u32 const dimension = blockIdx.y * fingerprint_threads_k + threadIdx.x;
    if (blockIdx.x >= merge_count || dimension >= dimensions) return;
    u32 const text_slot = merge_list[blockIdx.x];
    u32 const segments_of_text = segment_prefix[text_slot + 1] - segment_prefix[text_slot];
    fingerprint_state_t state = {fingerprint_skipped_k, 0};
    for (u32 s = 0; s < segments_of_text; ++s) {
        u64 const slot = ((u64)partial_prefix[text_slot] + s) * dimensions + dimension;
        fingerprint_state_t const partial = {partial_minimums[slot], partial_counts[slot]};
        fingerprint_merge(state, partial);
    }
    u32 const row = texts[text_slot].index;
    u32 *const hash_row = reinterpret_cast<u32 *>(reinterpret_cast<char *>(min_hashes) + (u64)row * min_hashes_stride);
    u32 *const count_row = reinterpret_cast<u32 *>(reinterpret_cast<char *>(min_counts) + (u64)row * min_counts_stride);
    fingerprint_export(state, hash_row + dimension, count_row + dimension);
}

} // namespace szs_hip

extern "C" int szs_hip_fingerprints(szs_string_ref_t const *texts, uint32_t texts_count, uint32_t const *segment_text,
                                    uint32_t const *segment_prefix, uint32_t const *partial_prefix,
                                    uint32_t total_segments, uint32_t const *merge_list,
                                    uint32_t merge_count, uint32_t dimensions, uint32_t const *widths,
                                    double const *multipliers, double const *modulos, double const *reciprocals,
                                    double const *complements, double *partial_minimums, uint32_t *partial_counts,
                                    uint32_t *min_hashes, uint64_t min_hashes_stride, uint32_t *min_counts,
                                    uint64_t min_counts_stride, uint32_t widest, void *stream) {
    using namespace szs_hip;
    if (!texts_count || !dimensions || !total_segments) return 0;
    hipStream_t const s = static_cast<hipStream_t>(stream);
    u32 const dimension_blocks = (dimensions + fingerprint_threads_k - 1) / fingerprint_threads_k;
    hipError_t error = hipSuccess;
    for (u32 first = 0; first < total_segments && error == hipSuccess; first += 1u << 30) { // keep each grid below 2^31 rows
        u32 const batch = total_segments - first < (1u << 30) ? total_segments - first : (1u << 30);
        if (widest <= fingerprint_max_width_k)
            hipLaunchKernelGGL(fingerprint_segments_kernel<true>, dim3(batch, dimension_blocks), dim3(fingerprint_threads_k), 0, s, texts,
                               first, segment_text, segment_prefix, partial_prefix, dimensions, widths, multipliers, modulos,
                               reciprocals, complements, partial_minimums, partial_counts, min_hashes, min_hashes_stride,
                               min_counts, min_counts_stride);
        else
            hipLaunchKernelGGL(fingerprint_segments_kernel<false>, dim3(batch, dimension_blocks), dim3(fingerprint_threads_k), 0, s, texts,
                               first, segment_text, segment_prefix, partial_prefix, dimensions, widths, multipliers, modulos,
                               reciprocals, complements, partial_minimums, partial_counts, min_hashes, min_hashes_stride,
                               min_counts, min_counts_stride);
        error = hipGetLastError();
    }
    if (error != hipSuccess || !merge_count) return (int)error;
    hipLaunchKernelGGL(fingerprint_merge_kernel, dim3(merge_count, dimension_blocks), dim3(fingerprint_threads_k), 0, s, texts,
                       merge_list, merge_count, segment_prefix, partial_prefix, dimensions, partial_minimums, partial_counts,
                       min_hashes, min_hashes_stride, min_counts, min_counts_stride);
    return (int)hipGetLastError();
}
