/*
 *  myers_core.hpp - the pieces of the bit-parallel unit-cost Levenshtein kernels that more than one translation unit needs:
 *  the LDS image of the match masks, the column update on a W-word bit-vector with one carry chain, and the strip column
 *  (Hyyro's block boundary) of the kernels that spread a pattern over lanes or strips.
 *
 *  Reference semantics: levenshtein_distance_myers<char, serial>, /root/reference/include/stringzillas/similarities/serial.hpp:2073-2314
 *  (block-based there; the full-width carry chain here computes the same DP column - tests pin this).
 *  Used by hip/lev_myers.hip (one launch per width group) and hip/myers_queue.hip (one persistent launch for every width).
 */
#pragma once
#include "device_common.hpp"

namespace szs_hip {

constexpr int byte_rows_k = 256;        // Peq rows of the byte kernels: one per byte value
constexpr int rune_slots_k = 512;       // Peq rows of the rune kernels: one per slot of the open-addressing rune table
constexpr u32 rune_slot_empty_k = ~0u;  // no decoded rune has this value (4-byte sequences top out below 2^21)

/** LDS image of Peq for a W-word pattern: 16-byte rows for W >= 3 (ds_read_b128), 8 for W = 2, 4 for W = 1.
 *  A row belongs to a byte value (byte kernels) or to a slot of the rune hash table (codepoint kernels). */
template <int words_, int rows_ = byte_rows_k>
struct peq_layout {
    static constexpr int chunk_words = words_ >= 3 ? 4 : words_;             // words fetched by one LDS read
    static constexpr int chunks = (words_ + chunk_words - 1) / chunk_words;  // LDS reads per text symbol
    static constexpr int total_dwords = chunks * rows_ * chunk_words;
    /** dword index of word `w` of the mask in row `row`: [chunk][row][word in chunk] */
    __device__ static constexpr int dword_index(int row, int w) {
        return ((w / chunk_words) * rows_ + row) * chunk_words + (w % chunk_words);
    }
};

/** Slot of `rune` in the workgroup's open-addressing table, or the empty slot its probe sequence ends on - whose Peq
 *  row is all zeros, exactly the match mask of a symbol the pattern does not contain. */
__device__ __forceinline__ u32 rune_slot_hash(u32 rune) { return (rune * 2654435761u) >> 23; }
__device__ __forceinline__ u32 find_rune_slot(u32 const *keys, u32 rune) {
    u32 slot = rune_slot_hash(rune);
    for (;;) {
        u32 const key = keys[slot];
        if (key == rune || key == rune_slot_empty_k) return slot;
        slot = (slot + 1) & (rune_slots_k - 1);
    }
}

template <int words_, int rows_>
__device__ __forceinline__ void load_match_masks(u32 const *peq, u32 symbol, u32 (&eq)[words_]) {
    using layout = peq_layout<words_, rows_>;
    if constexpr (layout::chunk_words == 4) {
        uint4 const *rows = reinterpret_cast<uint4 const *>(peq);
#pragma unroll
        for (int chunk = 0; chunk < layout::chunks; ++chunk) {
            uint4 const row = rows[chunk * rows_ + symbol];
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

/** Does `ref` still describe string `ref.index` of its tape?  (szs_ref_guard_t, hip/kernels.h.) */
__device__ __forceinline__ bool ref_is_current(szs_ref_guard_t const &guard, int side, szs_string_ref_t const &ref) {
    if (ref.index >= guard.side[side].count) return false;
    u64 from, to;
    if (guard.side[side].wide) {
        u64 const *offsets = static_cast<u64 const *>(guard.side[side].offsets);
        from = offsets[ref.index], to = offsets[(u64)ref.index + 1];
    }
    else {
        u32 const *offsets = static_cast<u32 const *>(guard.side[side].offsets);
        from = offsets[ref.index], to = offsets[(u64)ref.index + 1];
    }
    return to >= from && to - from == ref.length && guard.side[side].base + from == ref.address;
}

/** One column of the DP matrix: consumes the match masks of one text byte and updates the vertical delta vectors. */
template <int words_>
__device__ __forceinline__ void myers_column(u32 (&vp)[words_], u32 (&vn)[words_], u32 const (&eq)[words_]) {
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
}

/** One column of one strip; `hp_in` / `hn_in` are the deltas entering the strip's first row as 0 / 1 values; the bit pair
 *  leaving its last row comes back as hp | hn << 1. */
template <int words_>
__device__ __forceinline__ u32 myers_strip_column(u32 (&vp)[words_], u32 (&vn)[words_], u32 const (&eq)[words_], u32 hp_in,
                                                  u32 hn_in) {
    u32 carry = 0, hp_below = 0, hn_below = 0;
#pragma unroll
    for (int w = 0; w < words_; ++w) {
        u32 const xv = eq[w] | vn[w];
        u32 const eq_in = w == 0 ? (eq[w] | hn_in) : eq[w]; // a -1 entering from above acts like a match in the first row
        u32 carry_out;
        u32 const sum = __builtin_addc(eq_in & vp[w], vp[w], carry, &carry_out);
        carry = carry_out;
        u32 const d0 = (sum ^ vp[w]) | eq_in;
        u32 const hp = vn[w] | ~(d0 | vp[w]);
        u32 const hn = vp[w] & d0;
        u32 const hp_shifted = w == 0 ? ((hp << 1) | hp_in) : __builtin_amdgcn_alignbit(hp, hp_below, 31);
        u32 const hn_shifted = w == 0 ? ((hn << 1) | hn_in) : __builtin_amdgcn_alignbit(hn, hn_below, 31);
        hp_below = hp, hn_below = hn;
        vp[w] = hn_shifted | ~(xv | hp_shifted);
        vn[w] = hp_shifted & xv;
    }
    return (hp_below >> 31) | ((hn_below >> 31) << 1);
}

} // namespace szs_hip
