/*
 *  lev_myers.hip - unit-cost byte-level Levenshtein distances on gfx950, bit-parallel (Myers 1999 / Hyyro 2003).
 *
 *  Replaces, for the ROCm build, the reference's family of unit-cost kernels
 *      unit_myers_*_per_cuda_thread_/warp_   /root/reference/include/stringzillas/similarities/cuda.cuh:2389-2860
 *  and must return exactly what the reference's serial scorer returns
 *      levenshtein_distance_myers<char, serial>   .../similarities/serial.hpp:2073-2314.
 *
 *  MI355X-first design (not a translation of either of the above):
 *
 *  - One workgroup = one QUERY x 256 CANDIDATES; one candidate per lane.  The query is the Myers *pattern*: its
 *    match masks Peq[byte] are built once per workgroup in LDS and shared by all 256 lanes; every lane streams its
 *    own candidate as the *text*.  Work per lane-step is therefore pure register VALU plus one LDS gather.
 *  - 32-bit words, not 64-bit: CDNA4 has no 64-bit integer VALU, so the bit-vector is a W-word big integer whose
 *    `(Eq & VP) + VP` is ONE carry chain (`v_add_co_u32` / `v_addc_co_u32`) and whose shifts are `v_alignbit_b32`.
 *    The reference's CPU form is block-based with explicit HP/HN carries between 64-bit blocks (serial.hpp:2182-2204);
 *    the full-width form computes the same column of the DP matrix, hence the same distance (tests pin this).
 *  - The pattern is RIGHT-aligned in its W words: position i sits at bit (32 W - len + i).  The `pad` low bits are
 *    phantom rows initialised VP = VN = Eq = 0; they stay inert (HP = 1, HN = 0 forever) and hand the `+1 per column`
 *    top-row boundary into the first real bit through the ordinary shift.  The score bit is then ALWAYS bit 31 of
 *    the top word - no per-query variable shift in the inner loop - and one kernel instance serves every query
 *    whose length fits in W words (a zero-length query degenerates to `distance = candidate length` by itself).
 *  - Candidates are length-sorted by the host, so the 64 lanes of a wave finish within a few steps of each other;
 *    the step loop runs to the wave's longest text and shorter lanes only stop accumulating their score.
 *  - Results are written straight to results[query.index * stride + candidate.index] (+ mirror when symmetric).
 *
 *  Cost per text byte per lane: 13 VALU per word + ~7 -> ~0.45 VALU/cell at W = 4; LDS: one ds_read_b128 per
 *  four words.  HBM: tapes once + 8 B per result; see DESIGN.md section 5 for the roofline arithmetic.
 */
#include "device_common.hpp"

namespace szs_hip {

/** LDS image of Peq for a W-word pattern: 16-byte rows for W >= 3 (ds_read_b128), 8 for W = 2, 4 for W = 1. */
template <int words_>
struct peq_layout {
    static constexpr int chunk_words = words_ >= 3 ? 4 : words_;             // words fetched by one LDS read
    static constexpr int chunks = (words_ + chunk_words - 1) / chunk_words;  // LDS reads per text byte
    static constexpr int total_dwords = chunks * 256 * chunk_words;
    /** dword index of word `w` of the mask of byte `symbol`: [chunk][symbol][word in chunk] */
    __device__ static constexpr int dword_index(int symbol, int w) {
        return ((w / chunk_words) * 256 + symbol) * chunk_words + (w % chunk_words);
    }
};

template <int words_>
__device__ __forceinline__ void load_match_masks(u32 const *peq, u32 symbol, u32 (&eq)[words_]) {
    using layout = peq_layout<words_>;
    if constexpr (layout::chunk_words == 4) {
        uint4 const *rows = reinterpret_cast<uint4 const *>(peq);
#pragma unroll
        for (int chunk = 0; chunk < layout::chunks; ++chunk) {
            uint4 const row = rows[chunk * 256 + symbol];
            if (chunk * 4 + 0 < words_) eq[chunk * 4 + 0] = row.x;
            if (chunk * 4 + 1 < words_) eq[chunk * 4 + 1] = row.y;
            if (chunk * 4 + 2 < words_) eq[chunk * 4 + 2] = row.z;
            if (chunk * 4 + 3 < words_) eq[chunk * 4 + 3] = row.w;
        }
    }
    else if constexpr (layout::chunk_words == 2) {
        uint2 const row = reinterpret_cast<uint2 const *>(peq)[symbol];
        eq[0] = row.x, eq[1] = row.y;
    }
    else { eq[0] = peq[symbol]; }
}

/**
 *  One column of the DP matrix: consumes the match masks of one text byte, updates the vertical delta vectors and
 *  returns the pre-shift horizontal deltas of the TOP word (bit 31 = the last pattern row).
 */
template <int words_>
__device__ __forceinline__ void myers_column(u32 (&vp)[words_], u32 (&vn)[words_], u32 const (&eq)[words_], u32 &hp_top,
                                             u32 &hn_top) {
    u32 carry = 0, hp_below = 0, hn_below = 0;
#pragma unroll
    for (int w = 0; w < words_; ++w) {
        u32 const xv = eq[w] | vn[w];
        u32 carry_out;
        u32 const sum = __builtin_addc(eq[w] & vp[w], vp[w], carry, &carry_out); // one link of the W-word carry chain
        carry = carry_out;
        u32 const d0 = (sum ^ vp[w]) | eq[w];
        u32 const hp = vn[w] | ~(d0 | vp[w]);
        u32 const hn = vp[w] & d0;
        // Shift the horizontal deltas up by one row; bit 0 of word 0 takes the constant `+1` of DP row zero.
        u32 const hp_shifted = w == 0 ? ((hp << 1) | 1u) : __builtin_amdgcn_alignbit(hp, hp_below, 31);
        u32 const hn_shifted = w == 0 ? (hn << 1) : __builtin_amdgcn_alignbit(hn, hn_below, 31);
        hp_below = hp, hn_below = hn;
        vp[w] = hn_shifted | ~(xv | hp_shifted);
        vn[w] = hp_shifted & xv;
    }
    hp_top = hp_below, hn_top = hn_below;
}

/**
 *  @tparam words_        32-bit words of the pattern bit-vector; every query of the launch fits in 32 * words_ bytes.
 *  @tparam text_dwords_  text dwords consumed per loop iteration (4 bytes each): 4 for short patterns, 1 for long ones
 *                        to keep the unrolled body inside the instruction cache.
 */
template <int words_, int text_dwords_>
__global__ __launch_bounds__(256) void levenshtein_myers_kernel(szs_string_ref_t const *__restrict__ queries,
                                                                 szs_string_ref_t const *__restrict__ candidates,
                                                                 u32 candidates_count, u32 candidate_blocks,
                                                                 u64 *__restrict__ results, u64 results_row_stride,
                                                                 int symmetric) {
    using layout = peq_layout<words_>;
    __shared__ __attribute__((aligned(16))) u32 peq[layout::total_dwords];

    u32 const query_slot = blockIdx.x / candidate_blocks;
    u32 const candidate_block = blockIdx.x % candidate_blocks;
    szs_string_ref_t const query = queries[query_slot];
    u32 const query_length = query.length;
    u32 const pad = 32u * words_ - query_length; // phantom low rows

    // ---- Peq: zero, then scatter the pattern's bits (LDS atomics; a 128-byte query is 128 ORs per workgroup).
    for (int i = threadIdx.x; i < layout::total_dwords; i += 256) peq[i] = 0;
    __syncthreads();
    {
        u8 const *pattern = reinterpret_cast<u8 const *>(query.address);
        for (u32 i = threadIdx.x; i < query_length; i += 256) {
            u32 const position = pad + i;
            atomicOr(&peq[layout::dword_index(pattern[i], (int)(position >> 5))], 1u << (position & 31));
        }
    }
    __syncthreads();

    // ---- this lane's candidate
    u32 const candidate_slot = candidate_block * SZS_CANDIDATES_PER_WORKGROUP + threadIdx.x;
    bool live = candidate_slot < candidates_count;
    szs_string_ref_t candidate = {0, 0, 0};
    if (live) candidate = candidates[candidate_slot];
    if (symmetric && candidate.index > query.index) live = false; // upper triangle: mirrored from below
    u32 const text_length = live ? candidate.length : 0;
    u32 const longest_in_wave = wave_max_u32(text_length);

    u32 vp[words_], vn[words_];
#pragma unroll
    for (int w = 0; w < words_; ++w) {
        u32 const first_bit = 32u * w;
        vp[w] = first_bit >= pad ? ~0u : (first_bit + 32u <= pad ? 0u : (~0u << (pad - first_bit)));
        vn[w] = 0;
    }
    u32 distance = query_length;

    text_stream_t const text(candidate.address, text_length);
    u32 raw_low = text.raw(0);
    u32 raw_next[text_dwords_];
#pragma unroll
    for (int d = 0; d < text_dwords_; ++d) raw_next[d] = text.raw(1 + d);

    for (u32 column = 0, dword = 0; column < longest_in_wave; column += 4 * text_dwords_, dword += text_dwords_) {
        u32 symbols[text_dwords_];
        symbols[0] = text.splice(raw_low, raw_next[0]);
#pragma unroll
        for (int d = 1; d < text_dwords_; ++d) symbols[d] = text.splice(raw_next[d - 1], raw_next[d]);
        raw_low = raw_next[text_dwords_ - 1];
        // Issue the next iteration's loads now; they retire under the VALU work below.
#pragma unroll
        for (int d = 0; d < text_dwords_; ++d) raw_next[d] = text.raw(dword + text_dwords_ + 1 + d);

#pragma unroll
        for (int step = 0; step < 4 * text_dwords_; ++step) {
            u32 const symbol = (symbols[step / 4] >> (8 * (step % 4))) & 0xFFu;
            u32 eq[words_];
            load_match_masks<words_>(peq, symbol, eq);
            u32 hp_top, hn_top;
            myers_column<words_>(vp, vn, eq, hp_top, hn_top);
            u32 const delta = (hp_top >> 31) - (hn_top >> 31);
            distance += column + step < text_length ? delta : 0u;
        }
    }

    if (live) {
        results[(u64)query.index * results_row_stride + candidate.index] = distance;
        if (symmetric && candidate.index != query.index)
            results[(u64)candidate.index * results_row_stride + query.index] = distance;
    }
}

template <int words_>
static int launch_myers(szs_string_ref_t const *queries, u32 queries_count, szs_string_ref_t const *candidates,
                        u32 candidates_count, u64 *results, u64 stride, int symmetric, hipStream_t stream) {
    constexpr int text_dwords = words_ <= 8 ? 4 : 1;
    u32 const candidate_blocks = (candidates_count + SZS_CANDIDATES_PER_WORKGROUP - 1) / SZS_CANDIDATES_PER_WORKGROUP;
    // Keep each grid under 2^30 workgroups; enormous cross-products are cut along the query axis.
    u32 const queries_per_launch = candidate_blocks ? (1u << 30) / candidate_blocks : queries_count;
    for (u32 first = 0; first < queries_count; first += queries_per_launch) {
        u32 const batch = queries_count - first < queries_per_launch ? queries_count - first : queries_per_launch;
        hipLaunchKernelGGL((levenshtein_myers_kernel<words_, text_dwords>), dim3(batch * candidate_blocks), dim3(256), 0,
                           stream, queries + first, candidates, candidates_count, candidate_blocks, results, stride,
                           symmetric);
        hipError_t const error = hipGetLastError();
        if (error != hipSuccess) return (int)error;
    }
    return 0;
}

} // namespace szs_hip

extern "C" int szs_hip_levenshtein_myers(unsigned words, szs_string_ref_t const *queries, uint32_t queries_count,
                                         szs_string_ref_t const *candidates, uint32_t candidates_count,
                                         uint64_t *results, uint64_t results_row_stride, int symmetric, void *stream) {
    using namespace szs_hip;
    if (!queries_count || !candidates_count) return 0;
    hipStream_t const s = static_cast<hipStream_t>(stream);
#define SZS_MYERS_CASE(W)                                                                                              \
    case W: return launch_myers<W>(queries, queries_count, candidates, candidates_count, results, results_row_stride, symmetric, s);
    switch (words) {
        SZS_MYERS_CASE(1)
        SZS_MYERS_CASE(2)
        SZS_MYERS_CASE(3)
        SZS_MYERS_CASE(4)
        SZS_MYERS_CASE(5)
        SZS_MYERS_CASE(6)
        SZS_MYERS_CASE(7)
        SZS_MYERS_CASE(8)
        SZS_MYERS_CASE(10)
        SZS_MYERS_CASE(12)
        SZS_MYERS_CASE(16)
        SZS_MYERS_CASE(20)
        SZS_MYERS_CASE(24)
        SZS_MYERS_CASE(32)
        SZS_MYERS_CASE(48)
        SZS_MYERS_CASE(64)
    default: return (int)hipErrorInvalidValue;
    }
#undef SZS_MYERS_CASE
}

/** The word counts the launcher has instances for, ascending; the host rounds each query up to the next one. */
extern "C" unsigned szs_hip_levenshtein_myers_round_words(unsigned words) {
    static unsigned const steps[] = {1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 16, 20, 24, 32, 48, 64};
    for (unsigned i = 0; i < sizeof(steps) / sizeof(steps[0]); ++i)
        if (words <= steps[i]) return steps[i];
    return 0; /* too long for the bit-parallel kernel */
}
