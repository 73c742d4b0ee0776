/*
 *  utf8.hip - transcoding UTF-8 tapes to UTF-32 on the device, the front end of the codepoint-level Levenshtein engine.
 *
 *  Replaces, for the ROCm build, the reference's per-pair transcoding
 *      levenshtein_distance_utf8::operator()   /root/reference/include/stringzillas/similarities/serial.hpp:2825-2837
 *      decode_utf8_rune / rune-offset index    .../similarities/cuda.cuh:3162-3175,3145-3703
 *  with ONE pass per call: a cross-product scores every string against many others, so each string is decoded once into
 *  a UTF-32 scratch tape and the scoring kernels stream fixed-width runes (one aligned dword per DP column) instead of
 *  re-decoding variable-width sequences Q x C times.
 *
 *  Value contract = `sz_rune_decode_unchecked` (include/stringzilla/utf8_runes/serial.h:111-124): the sequence length
 *  comes from the lead byte alone, continuation bytes contribute their low six bits unvalidated, a stray continuation
 *  byte is a one-byte rune, nothing is ever rejected.  Bytes missing from a sequence truncated by the end of the string
 *  read as zero (the reference over-reads there).
 *
 *  Layout: string i's runes start at `rune_starts[i]` (the host passes the prefix sum of BYTE lengths: a string never
 *  has more runes than bytes, so slots cannot collide and no device-side scan is needed).
 */
#include "device_common.hpp"

namespace szs_hip {

/**
 *  One string per thread; strings are independent and the pass is O(bytes) next to O(Q C len^2 / 32).  What costs here is
 *  LATENCY, not work: round 1 read the string a byte at a time, every read dependent on the previous rune's length - one
 *  global-memory round trip per rune, 0.9 ms for config 5's 2 KB strings.  Now a thread walks its string in aligned 16-byte
 *  chunks held in registers, the next chunk in flight while the current one is decoded: a round trip per 16 bytes, hidden.
 *  Only chunks that overlap the string are ever loaded (same 16-byte line as a byte the caller owns).
 */
__global__ __launch_bounds__(256) void utf8_transcode_kernel(szs_string_ref_t const *__restrict__ strings, u32 count,
                                                             u64 const *__restrict__ rune_starts,
                                                             u32 *__restrict__ runes, u32 *__restrict__ rune_counts,
                                                             u32 *__restrict__ any_multibyte) {
    u32 const i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    u64 const address = strings[i].address;
    u32 const length = strings[i].length;
    u32 *const out = runes + rune_starts[i];
    uint4 const *const lines = reinterpret_cast<uint4 const *>(address & ~(u64)15);
    u32 const skew = (u32)(address & 15);                          // string byte b is stream byte skew + b
    u32 const line_count = length ? (skew + length + 15) / 16 : 0; // lines that hold string bytes
    uint4 const zero = make_uint4(0, 0, 0, 0);
    uint4 current = line_count ? lines[0] : zero, next = line_count > 1 ? lines[1] : zero;
    u32 line = 0, produced = 0, multibyte = 0;
    // the four stream bytes starting at stream position `at`, which lies in the current line
    auto four_bytes = [&](u32 at) -> u32 {
        u32 const word = (at >> 2) & 3u;
        u32 const low = word == 0 ? current.x : word == 1 ? current.y : word == 2 ? current.z : current.w;
        u32 const high = word == 0 ? current.y : word == 1 ? current.z : word == 2 ? current.w : next.x;
        return __builtin_amdgcn_alignbyte(high, low, at & 3u);
    };
    for (u32 progress = 0; progress < length;) {
        u32 const at = skew + progress;
        if ((at >> 4) != line) { // the lead byte lies in the next line: it becomes the current one, its successor is fetched
            ++line;
            current = next;
            next = line + 1 < line_count ? lines[line + 1] : zero;
        }
        u32 const bytes = four_bytes(at);
        u32 const lead = bytes & 0xFFu;
        u32 const sequence = 1u + (lead >= 0xC0u) + (lead >= 0xE0u) + (lead >= 0xF0u);
        u32 tail[3];
#pragma unroll
        for (u32 k = 1; k < 4; ++k) tail[k - 1] = k < sequence && progress + k < length ? (bytes >> (8 * k)) & 0x3Fu : 0u;
        u32 rune = lead;
        if (sequence == 2) rune = (lead & 0x1Fu) << 6 | tail[0];
        if (sequence == 3) rune = (lead & 0x0Fu) << 12 | tail[0] << 6 | tail[1];
        if (sequence == 4) rune = (lead & 0x07u) << 18 | tail[0] << 12 | tail[1] << 6 | tail[2];
        out[produced++] = rune;
        progress += sequence;
        multibyte |= lead >= 0x80u; // any byte >= 0x80 takes the pair off the reference's ASCII shortcut (serial.hpp:2809)
    }
    rune_counts[i] = produced;
    if (multibyte) atomicOr(any_multibyte, 1u);
}

} // namespace szs_hip

extern "C" int szs_hip_utf8_transcode(szs_string_ref_t const *strings, uint32_t count, uint64_t const *rune_starts,
                                      uint32_t *runes, uint32_t *rune_counts, uint32_t *any_multibyte, void *stream) {
    using namespace szs_hip;
    if (!count) return 0;
    hipLaunchKernelGGL(utf8_transcode_kernel, dim3((count + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream),
                       strings, count, rune_starts, runes, rune_counts, any_multibyte);
    return (int)hipGetLastError();
}
